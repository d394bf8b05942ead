// Standalone check + timing harness for lstc_gemm (links liblstc_hip.so, no torch).
//   tools/gemm_check            correctness on odd shapes (vs a double-precision host reference) + timing table
//   tools/gemm_check time       timing only
//   tools/gemm_check check      correctness only (tests/test_hip_parity.py runs this as a -m gpu test)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "lstc_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static std::vector<float> rnd(size_t n, unsigned seed) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::vector<float> v(n);
    for (auto& x : v) x = d(g);
    return v;
}

struct Case { int M, N, K, tA, tB, flags, variant, split, dtype = 0, alignc = 0; };   // alignc: C / residual / mask leading dims multiples of 4 (16-B epilogue path)

static float bf16_round(float x) {   // RNE to bfloat16, back to float (host model of the kernel's operand rounding)
    uint32_t u; memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    float y; memcpy(&y, &u, 4); return y;
}

static int check(const Case& c) {
    const int M = c.M, N = c.N, K = c.K;
    const bool pad = (c.variant % 2 == 0) && c.variant != 4 && !(c.dtype == 0 && c.variant == 12 && c.alignc);     // padded lds exercise the scalar-load path
    const int lda = (c.tA ? M : K) + (pad ? 3 : 0);
    const int ldb = (c.tB ? K : N) + (pad ? 1 : 0);
    const int ldc = c.alignc ? N + 4 : N + 2, ldr = c.alignc ? N + 8 : N + 1, ldm = N;
    const size_t na = (size_t)(c.tA ? K : M) * lda, nb = (size_t)(c.tB ? N : K) * ldb;
    auto hA = rnd(na, 1), hB = rnd(nb, 2), hbias = rnd(N, 3), hres = rnd((size_t)M * ldr, 4), hmask = rnd((size_t)M * ldm, 5);
    std::vector<float> hC((size_t)M * ldc, 0.5f);
    float *dA, *dB, *dC, *dbias, *dres, *dmask;
    CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dB, nb * 4)); CK(hipMalloc(&dC, hC.size() * 4));
    CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dres, hres.size() * 4)); CK(hipMalloc(&dmask, hmask.size() * 4));
    CK(hipMemcpy(dA, hA.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dres, hres.data(), hres.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmask, hmask.data(), hmask.size() * 4, hipMemcpyHostToDevice));
    if (c.split > 1) std::fill(hC.begin(), hC.end(), 0.f);
    CK(hipMemcpy(dC, hC.data(), hC.size() * 4, hipMemcpyHostToDevice));
    LstcGemmDesc d; memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc; d.transA = c.tA; d.transB = c.tB;
    d.dtype = c.dtype; d.flags = c.flags; d.alpha = 0.75f; d.dropout_p = 0.f; d.ldr = ldr; d.ld_relu = ldm;
    d.split_k = c.split; d.variant = c.variant; d.A = dA; d.B = dB; d.C = dC; d.bias = dbias; d.residual = dres; d.relu_src = dmask;
    void *pA = nullptr, *pB = nullptr;
    const bool p1 = c.dtype == LSTC_BF16P;
    auto packf = [&](const float* s_, int64_t r_, int64_t k_, int64_t ld_, int km_, void* d_) {
        return p1 ? lstc_pack1(s_, r_, k_, ld_, km_, d_, nullptr) : lstc_pack3(s_, r_, k_, ld_, km_, d_, nullptr); };
    auto packb = [&](int64_t r_, int64_t k_) { return p1 ? lstc_pack1_bytes(r_, k_) : lstc_pack3_bytes(r_, k_); };
    if (c.dtype == LSTC_F32X3 || p1) {     // operands go through lstc_pack3 / lstc_pack1: A as [M, K], B as [N, K]
        const bool trc = c.variant >= 7 && c.variant <= 9;
        CK(hipMalloc(&pA, trc ? packb(K, M) : packb(M, K))); CK(hipMalloc(&pB, trc ? packb(K, N) : packb(N, K)));
        CK(hipMemset(pA, 0xff, trc ? packb(K, M) : packb(M, K))); CK(hipMemset(pB, 0xff, trc ? packb(K, N) : packb(N, K)));   // NaN-fill: padding must be written by the pack
        int r1, r2;
        if (c.variant >= 7 && c.variant <= 9) {            // weight-gradient form (7: three-stage kernel, 8: 256x128-tile kernel, 9: two-stage kernel): packs of the SOURCES [K, M], [K, N] + transposed reads
            r1 = packf(dA, K, M, lda, 0, pA); r2 = packf(dB, K, N, ldb, 0, pB);
            d.transA = 1; d.transB = 0; d.variant = p1 ? 0 : c.variant == 8 ? 2 : c.variant == 9 ? 3 : 1;   /* 7 -> three-stage kernel */
        } else {
            r1 = packf(dA, M, K, lda, c.tA ? 1 : 0, pA); r2 = packf(dB, N, K, ldb, c.tB ? 0 : 1, pB);
            d.transA = 0; d.transB = 1; if (p1) d.variant = 0;
        }
        if (r1 || r2) { printf("lstc_pack rc=%d/%d\n", r1, r2); return 1; }
        d.A = pA; d.B = pB;
    }
    int rc = lstc_gemm(&d, nullptr);
    if (rc) { printf("lstc_gemm rc=%d (%s)\n", rc, lstc_strerror(rc)); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> out(hC.size());
    CK(hipMemcpy(out.data(), dC, out.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) {
                float af = c.tA ? hA[(size_t)k * lda + m] : hA[(size_t)m * lda + k];
                float bf = c.tB ? hB[(size_t)n * ldb + k] : hB[(size_t)k * ldb + n];
                if (c.dtype == LSTC_BF16 || c.dtype == LSTC_BF16P) { af = bf16_round(af); bf = bf16_round(bf); }
                s += (double)af * (double)bf;
            }
            double v = s * 0.75;
            if (c.split <= 1) {
                if (c.flags & LSTC_EPI_BIAS) v += hbias[n];
                if (c.flags & LSTC_EPI_RELU) v = v > 0 ? v : 0;
                if (c.flags & LSTC_EPI_RESIDUAL) v += hres[(size_t)m * ldr + n];
                if (c.flags & LSTC_EPI_RELU_MASK) v = hmask[(size_t)m * ldm + n] > 0 ? v : 0;
                if (c.flags & LSTC_EPI_ACCUM) v += 0.5;
            }
            maxerr = std::fmax(maxerr, std::fabs(v - out[(size_t)m * ldc + n]));
        }
    // padding columns of C must be untouched
    bool pad_ok = true;
    for (int m = 0; m < M; ++m)
        for (int n = N; n < ldc; ++n) pad_ok &= out[(size_t)m * ldc + n] == (c.split > 1 ? 0.f : 0.5f);
    // f32x3 is held to the f32 kernels' bound (operands in [-1, 1): |sum| <~ sqrt(K))
    const bool ok = maxerr < 2e-4 * std::sqrt((double)K) * (c.dtype == LSTC_F32X3 ? 0.02 : 1.0) && pad_ok;
    if (pA) hipFree(pA);
    if (pB) hipFree(pB);
    printf("%s %s M=%d N=%d K=%d tA=%d tB=%d flags=%d var=%d split=%d maxerr=%.3g pad_ok=%d\n", ok ? "PASS" : "FAIL",
           c.dtype == LSTC_F32X3 ? "f32x3" : c.dtype == LSTC_BF16P ? "bf16p" : c.dtype ? "bf16c" : "f32", M, N, K, c.tA, c.tB, c.flags, c.variant, c.split, maxerr, (int)pad_ok);
    fflush(stdout);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dbias); hipFree(dres); hipFree(dmask);
    return ok ? 0 : 1;
}

static void timeit(int M, int N, int K, int tA, int tB, int variant, int split, int flags, int iters, int pad_a = 0, int pad_b = 0, int dtype = 0) {
    const int v15 = variant & 15;          // the tile / form selector; the bits above it select tuning ablations (LSTC_TUNING builds)
    const int lda = (tA ? M : K) + pad_a, ldb = (tB ? K : N) + pad_b;
    const size_t na = (size_t)(tA ? K : M) * lda, nb = (size_t)(tB ? N : K) * ldb, nc = (size_t)M * N;
    float *dA, *dB, *dC, *dbias;
    CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dB, nb * 4)); CK(hipMalloc(&dC, nc * 4)); CK(hipMalloc(&dbias, N * 4));
    {   // random fill (DVFS: never time on zeros)
        std::vector<float> h = rnd(1 << 22, 7);
        for (size_t o = 0; o < na; o += h.size()) CK(hipMemcpy(dA + o, h.data(), std::min(h.size(), na - o) * 4, hipMemcpyHostToDevice));
        for (size_t o = 0; o < nb; o += h.size()) CK(hipMemcpy(dB + o, h.data(), std::min(h.size(), nb - o) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbias, h.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0, nc * 4));
    }
    LstcGemmDesc d; memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = N; d.transA = tA; d.transB = tB; d.dtype = dtype;
    d.flags = flags; d.alpha = 1.f; d.split_k = split; d.variant = variant; d.A = dA; d.B = dB; d.C = dC; d.bias = dbias;
    d.residual = dC; d.ldr = N; d.dropout_p = 0.1f; d.dropout_seed = 5;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    void *pA = nullptr, *pB = nullptr;
    const bool p1 = dtype == LSTC_BF16P;
    auto packf = [&](const float* s_, int64_t r_, int64_t k_, int64_t ld_, int km_, void* d_) {
        return p1 ? lstc_pack1(s_, r_, k_, ld_, km_, d_, nullptr) : lstc_pack3(s_, r_, k_, ld_, km_, d_, nullptr); };
    if (dtype == LSTC_F32X3 || p1) {
        const bool trc = v15 >= 7 && v15 <= 9;
        CK(hipMalloc(&pA, p1 ? (trc ? lstc_pack1_bytes(K, M) : lstc_pack1_bytes(M, K)) : lstc_pack3_bytes(M, K)));
        CK(hipMalloc(&pB, p1 ? (trc ? lstc_pack1_bytes(K, N) : lstc_pack1_bytes(N, K)) : lstc_pack3_bytes(N, K)));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, nullptr));
            if (v15 >= 7 && v15 <= 9) packf(dA, K, M, lda, 0, pA); else packf(dA, M, K, lda, tA ? 1 : 0, pA);
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float msa; CK(hipEventElapsedTime(&msa, e0, e1));
            CK(hipEventRecord(e0, nullptr));
            if (v15 >= 7 && v15 <= 9) packf(dB, K, N, ldb, 0, pB); else packf(dB, N, K, ldb, tB ? 0 : 1, pB);
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float msb; CK(hipEventElapsedTime(&msb, e0, e1));
            if (rep) printf("PACK A [%d x %d]%s %.3f ms (%.2f TB/s read + write)   B [%d x %d]%s %.3f ms\n", M, K, tA ? " k-major" : "", msa,
                            (p1 ? 6.0 : 12.0) * M * K / (msa * 1e-3) / 1e12, N, K, tB ? "" : " k-major", msb);
        }
        d.A = pA; d.B = pB;
        if (v15 >= 7 && v15 <= 9) { d.transA = 1; d.transB = 0; d.variant = p1 ? (variant & ~15) : v15 == 8 ? 2 : v15 == 9 ? 3 : 1; } else { d.transA = 0; d.transB = 1; if (p1) d.variant = variant & ~15; }
    }
    void *pC = nullptr, *pR = nullptr;
    if (p1 && (flags & LSTC_EPI_OUT_PACK)) {                // timing of the packed-output / packed-residual epilogues (bf16 activation stream)
        CK(hipMalloc(&pC, lstc_pack1_bytes(M, N)));
        d.C = pC;
        if (flags & LSTC_EPI_RESIDUAL_PACK) {
            CK(hipMalloc(&pR, lstc_pack1_bytes(M, N)));
            lstc_pack1(dC, M, N, N, 0, pR, nullptr);
            d.residual = pR;
        }
    }
#ifdef LSTC_TUNING
    unsigned long long* dstamp = nullptr;
    const bool stamps = p1 && ((variant >> 4) & 0x2000) && !(flags & LSTC_EPI_RELU_MASK);
    if (stamps) { CK(hipMalloc(&dstamp, 1024 * 100 * 8)); CK(hipMemset(dstamp, 0, 1024 * 100 * 8)); d.relu_src = (const float*)dstamp; }
#endif
    for (int i = 0; i < 6; ++i) { int rc = lstc_gemm(&d, nullptr); if (rc) { printf("rc=%d\n", rc); return; } }   // clock ramp
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) lstc_gemm(&d, nullptr);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double tf = 2.0 * M * N * (double)K / (ms * 1e-3) / 1e12;
    const double peak = (dtype == LSTC_BF16 || dtype == LSTC_BF16P) ? 2500.0 : 157.3;
    printf("TIME %s M=%6d N=%5d K=%6d tA=%d tB=%d var=%d split=%d flags=%3d pad=%d,%d : %8.3f ms  %7.2f TFLOP/s (%.1f%% of %.1f)\n", dtype == LSTC_F32X3 ? "f32x3" : dtype == LSTC_BF16P ? "bf16p" : dtype ? "bf16c" : "f32", M, N, K, tA, tB,
           variant, split, flags, pad_a, pad_b, ms, tf, 100.0 * tf / peak, peak);
#ifdef LSTC_TUNING
    if (stamps) {
        // per persistent workgroup: [0..3] = memtime / memrealtime at start and end, then per item 4 stamps (item start, K step 0 landed,
        // K loop done, epilogue done).  Shader cycles -> us through the workgroup's own cycles per 100-MHz tick.
        std::vector<unsigned long long> h(1024 * 100);
        CK(hipMemcpy(h.data(), dstamp, h.size() * 8, hipMemcpyDeviceToHost));
        double sum[3][4] = {{0}}, cnt[3] = {0, 0, 0}, span = 0, clk = 0; int nb = 0, maxit = 0;
        double xspan[8] = {0}, xclk[8] = {0}, xend[8] = {0}, xstart[8] = {0}; int xn[8] = {0};
        double smin = 1e30, smax = 0, s13 = 0, s12 = 0; int n13 = 0, n12 = 0;
        unsigned long long r0 = ~0ull, r1 = 0;
        for (int b = 0; b < 1024; ++b) {
            const unsigned long long* s = &h[(size_t)b * 100];
            if (!s[0] || s[3] <= s[1]) continue;
            const double cyc_per_us = (double)(s[2] - s[0]) / ((double)(s[3] - s[1]) / 100.0);
            int n = 0; while (n < 24 && s[4 + 4 * n + 3]) ++n;
            if (n == 0) continue;
            ++nb; clk += cyc_per_us; span += (double)(s[2] - s[0]) / cyc_per_us; maxit = std::max(maxit, n);
            {   // by XCD (blockIdx % 8 share one under round-robin placement) and by item count; realtime (100 MHz) brackets of the launch
                const double sp = (double)(s[2] - s[0]) / cyc_per_us;
                const int x = b & 7;
                xspan[x] += sp; xclk[x] += cyc_per_us; ++xn[x];
                smin = std::min(smin, sp); smax = std::max(smax, sp);
                if (n >= 13) { s13 += sp; ++n13; } else { s12 += sp; ++n12; }
                r0 = std::min(r0, s[1]); r1 = std::max(r1, s[3]);
                xstart[x] += (double)s[1]; xend[x] += (double)s[3];
            }
            for (int i = 0; i < n; ++i) {
                const unsigned long long* t = s + 4 + 4 * i;
                const int cls = i == 0 ? 0 : (i == n - 1 ? 2 : 1);
                sum[cls][0] += (double)(t[1] - t[0]) / cyc_per_us; sum[cls][1] += (double)(t[2] - t[1]) / cyc_per_us;
                sum[cls][2] += (double)(t[3] - t[2]) / cyc_per_us;
                sum[cls][3] += i + 1 < n ? (double)(t[4] - t[3]) / cyc_per_us : 0.0;
                cnt[cls] += 1;
            }
        }
        printf("STAMPS %d workgroups, up to %d items each, in-kernel clock %.0f MHz, mean workgroup span %.1f us (launch %.1f us)\n", nb, maxit,
               nb ? clk / nb : 0.0, nb ? span / nb : 0.0, ms * 1e3);
        printf("STAMPS workgroup spans: min %.1f max %.1f us; %d workgroups with >= 13 items mean %.1f us, %d with fewer mean %.1f us; first start -> last end %.1f us\n",
               smin, smax, n13, n13 ? s13 / n13 : 0.0, n12, n12 ? s12 / n12 : 0.0, (double)(r1 - r0) / 100.0);
        printf("STAMPS by blockIdx %% 8 (one XCD each): ");
        for (int x = 0; x < 8; ++x) if (xn[x]) printf("[%d: %.0f MHz, span %.1f us, ends +%.1f us] ", x, xclk[x] / xn[x], xspan[x] / xn[x], (xend[x] / xn[x] - (double)r0) / 100.0);
        printf("\n");
        const char* nm[3] = {"first item ", "middle items", "last item  "};
        for (int c = 0; c < 3; ++c)
            if (cnt[c] > 0)
                printf("STAMPS %s (%6.0f): wait for K step 0 %6.2f us | K loop %6.2f us | next-item head + epilogue %6.2f us | to next item %5.2f us\n", nm[c], cnt[c],
                       sum[c][0] / cnt[c], sum[c][1] / cnt[c], sum[c][2] / cnt[c], sum[c][3] / cnt[c]);
        hipFree(dstamp);
    }
#endif
    if (pA) hipFree(pA);
    if (pB) hipFree(pB);
    if (pC) hipFree(pC);
    if (pR) hipFree(pR);
    fflush(stdout);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dbias);
}

int main(int argc, char** argv) {
    if (argc >= 10 && !strcmp(argv[1], "one")) {   // one M N K tA tB variant split flags iters
        timeit(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]),
               atoi(argv[9]), argc > 10 ? atoi(argv[10]) : 3, argc > 11 ? atoi(argv[11]) : 0, argc > 12 ? atoi(argv[12]) : 0, argc > 13 ? atoi(argv[13]) : 0);
        return 0;
    }
    if (argc >= 12 && !strcmp(argv[1], "case")) {   // case M N K tA tB flags variant split dtype alignc
        return check({atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), atoi(argv[9]),
                      atoi(argv[10]), atoi(argv[11])});
    }
    const bool time_only = argc > 1 && !strcmp(argv[1], "time");
    const bool check_only = argc > 1 && !strcmp(argv[1], "check");
    int fails = 0;
    if (!time_only) {
        const int ALL = LSTC_EPI_BIAS | LSTC_EPI_RELU | LSTC_EPI_RESIDUAL | LSTC_EPI_RELU_MASK | LSTC_EPI_ACCUM;
        // production library: 0 (default; padded lds -> the PIPE 3 scalar-load fallback), 8 (PIPE 3 itself), 4 (PIPE 5, aligned), 12 (persistent;
        // padded lds -> its fallback); `make tuning` builds also hold 2 / 6 (256x128 tiles) and 10 (LDS-DMA staging)
#ifdef LSTC_TUNING
        for (int variant : {0, 8, 4, 12, 2, 6, 10}) {
#else
        for (int variant : {0, 8, 4, 12}) {
#endif
            fails += check({300, 200, 100, 0, 1, 0, variant, 1});
            fails += check({257, 131, 67, 0, 1, ALL, variant, 1});
            fails += check({300, 200, 100, 0, 0, LSTC_EPI_RELU_MASK, variant, 1});
            fails += check({130, 260, 515, 1, 0, 0, variant, 1});
            fails += check({130, 260, 515, 1, 0, 0, variant, 3});
            fails += check({64, 1, 32, 0, 1, LSTC_EPI_BIAS, variant, 1});
        }
#ifdef LSTC_TUNING
        // aligned (vector-path) tuning variants: variant id odd -> no ld padding in check()
        for (int variant : {1, 3, 5, 7, 9}) {
            fails += check({256, 256, 128, 0, 1, 0, variant, 1});
            fails += check({256, 256, 128, 0, 0, 0, variant, 1});
            fails += check({256, 256, 128, 1, 0, 0, variant, 1});
            fails += check({384, 132, 260, 1, 0, 0, variant, 2});
        }
#endif
        // the persistent walk on aligned operands (alignc: 16-B epilogue path - what variant 12 needs to run its own kernel)
        fails += check({1024, 512, 256, 0, 1, LSTC_EPI_BIAS | LSTC_EPI_RELU, 12, 1, 0, 1});
        fails += check({1024, 512, 256, 0, 0, LSTC_EPI_RESIDUAL, 12, 1, 0, 1});
        // LDS-DMA variant (10): aligned leading dims (odd variant id would pad; 10 is even -> use sizes whose padded lds
        // stay multiples of 4 is impossible, so these go through variant 11 = same kernel, no padding)
        for (int tb : {1, 0}) {
            fails += check({256, 256, 128, 0, tb, 0, 11, 1});
            fails += check({300, 200, 96, 0, tb, ALL, 11, 1});
            fails += check({1000, 260, 320, 0, tb, LSTC_EPI_BIAS | LSTC_EPI_RELU, 11, 1});
        }
        fails += check({256, 256, 128, 1, 0, 0, 11, 1});
        fails += check({300, 200, 96, 1, 0, 0, 11, 1});
        fails += check({384, 132, 512, 1, 0, 0, 11, 3});
        for (int tb : {1, 0}) {      // buffer-load pipeline (variant 4), aligned operands, odd tile counts, K tails
            fails += check({256, 256, 128, 0, tb, 0, 4, 1});
            fails += check({300, 200, 100, 0, tb, ALL, 4, 1});
            fails += check({1000, 260, 1000, 0, tb, LSTC_EPI_BIAS | LSTC_EPI_RELU, 4, 1});
        }
        fails += check({256, 256, 640, 1, 0, 0, 4, 1});
        fails += check({300, 200, 1000, 1, 0, 0, 4, 1});
        fails += check({384, 132, 2048, 1, 0, 0, 4, 3});
        const int ALLB = LSTC_EPI_BIAS | LSTC_EPI_RELU | LSTC_EPI_RESIDUAL | LSTC_EPI_RELU_MASK | LSTC_EPI_ACCUM;
        for (int variant : {0, 1, 2}) {        // bf16-compute kernel; odd variant id -> aligned (vector) loads
            fails += check({300, 200, 100, 0, 1, 0, variant, 1, LSTC_BF16});
            fails += check({257, 131, 67, 0, 1, ALLB, variant, 1, LSTC_BF16});
            fails += check({300, 200, 132, 0, 0, LSTC_EPI_RELU_MASK, variant, 1, LSTC_BF16});
            fails += check({130, 260, 515, 1, 0, 0, variant, 1, LSTC_BF16});
            fails += check({132, 260, 512, 1, 0, 0, variant, 3, LSTC_BF16});
            fails += check({64, 1, 32, 0, 1, LSTC_EPI_BIAS, variant, 1, LSTC_BF16});
        }
        for (int split : {1, 3}) {             // packed f32x3 kernel: every layout goes through lstc_pack3
            fails += check({300, 200, 100, 0, 1, 0, 0, 1, LSTC_F32X3});
            fails += check({257, 131, 67, 0, 1, ALLB, 1, 1, LSTC_F32X3});
            fails += check({300, 200, 132, 0, 0, LSTC_EPI_RELU_MASK, 0, 1, LSTC_F32X3});
            fails += check({130, 260, 515, 1, 0, 0, 1, split, LSTC_F32X3});
            fails += check({64, 1, 32, 0, 1, LSTC_EPI_BIAS, 0, 1, LSTC_F32X3});
            fails += check({256, 384, 32 * (3 + split), 0, 1, 0, 1, 1, LSTC_F32X3});
            fails += check({128, 128, 64, 0, 1, 0, 1, 1, LSTC_F32X3});
            fails += check({500, 260, 1000, 1, 0, 0, 1, split + 1, LSTC_F32X3});
            fails += check({256, 128, 384, 1, 0, 0, 7, split, LSTC_F32X3});
            fails += check({128, 384, 1152, 1, 0, 0, 7, split + 1, LSTC_F32X3});
            fails += check({256, 384, 640, 1, 0, 0, 9, split, LSTC_F32X3});                  // 2-stage kernel, TR
            fails += check({512, 256, 640, 1, 0, 0, 8, split, LSTC_F32X3});                  // 256x128-tile kernel, TR
            fails += check({300, 520, 100 + 32 * split, 0, 1, ALLB, 2, 1, LSTC_F32X3});     // 256x128-tile kernel, ragged NT
            fails += check({512, 200, 96, 0, 0, LSTC_EPI_RELU_MASK, 2, 1, LSTC_F32X3});
            fails += check({260, 130, 515, 1, 0, 0, 2, split, LSTC_F32X3});
            fails += check({300, 520, 100 + 32 * split, 0, 1, ALLB, 3, 1, LSTC_F32X3});     // 2-stage kernel, ragged NT
            fails += check({130, 260, 515 + 32 * split, 1, 0, 0, 3, split, LSTC_F32X3});
        }
        for (int split : {1, 3}) {             // packed bf16 kernel (256x256x64 tiles, 8 waves): every layout goes through lstc_pack1
            fails += check({300, 200, 100, 0, 1, 0, 0, 1, LSTC_BF16P});
            fails += check({257, 131, 67, 0, 1, ALLB, 1, 1, LSTC_BF16P});
            fails += check({300, 200, 132, 0, 0, LSTC_EPI_RELU_MASK, 0, 1, LSTC_BF16P});
            fails += check({130, 260, 515, 1, 0, 0, 1, split, LSTC_BF16P});
            fails += check({64, 1, 32, 0, 1, LSTC_EPI_BIAS, 0, 1, LSTC_BF16P});
            fails += check({256, 384, 64 * (3 + split), 0, 1, 0, 1, 1, LSTC_BF16P});
            fails += check({512, 512, 64, 0, 1, 0, 1, 1, LSTC_BF16P});
            fails += check({600, 520, 1000, 0, 1, ALLB, 1, 1, LSTC_BF16P});
            fails += check({500, 260, 1000, 1, 0, 0, 1, split + 1, LSTC_BF16P});
            fails += check({256, 256, 384, 1, 0, 0, 7, split, LSTC_BF16P});                 // TR form (weight gradient)
            fails += check({512, 256, 1152, 1, 0, 0, 7, split + 1, LSTC_BF16P});
            fails += check({300, 523, 640, 1, 0, 0, 7, split, LSTC_BF16P});                 // TR, ragged feature counts
            fails += check({96, 40, 128, 1, 0, 0, 7, 1, LSTC_BF16P});
            fails += check({600, 520, 1000, 0, 1, ALLB, 1, 1, LSTC_BF16P, 1});              // 16-B (quad-transposed) epilogue, every flag
            fails += check({257, 132, 67, 0, 1, LSTC_EPI_BIAS | LSTC_EPI_RELU, 1, 1, LSTC_BF16P, 1});
            fails += check({300, 200, 132, 0, 0, LSTC_EPI_RELU_MASK | LSTC_EPI_ACCUM, 0, 1, LSTC_BF16P, 1});
            fails += check({512, 256, 1152, 1, 0, 0, 7, 1, LSTC_BF16P, 1});
            // hand-counted epilogue (whole 256 x 256 tiles, 16-B accesses), one operand stream at a time; 264 tiles: items with a successor
            fails += check({512, 256, 256, 0, 1, LSTC_EPI_RELU_MASK, 1, 1, LSTC_BF16P, 1});
            fails += check({512, 512, 320, 0, 1, LSTC_EPI_BIAS | LSTC_EPI_RELU, 1, 1, LSTC_BF16P, 1});
            fails += check({1024, 256, 192, 0, 1, LSTC_EPI_BIAS | LSTC_EPI_DROPOUT | LSTC_EPI_RESIDUAL, 1, 1, LSTC_BF16P, 1});
            fails += check({8448, 2048, 64, 0, 1, LSTC_EPI_RESIDUAL, 1, 1, LSTC_BF16P, 1});
            fails += check({8448, 2048, 128, 0, 1, LSTC_EPI_RELU_MASK | LSTC_EPI_BIAS, 1, 1, LSTC_BF16P, 1});
        }
        printf("%s: %d failing cases\n", fails ? "FAILED" : "ALL PASS", fails);
    }
    if (check_only) return fails;
    // LTN headline shapes: tokens M = 2048*49 = 100352, d = 2048, Hd = 2048, F = 4096.  A smaller M (25088)
    // is timed first to keep the table quick; TFLOP/s is what matters.
    const int Mtok = 100352;
#ifdef LSTC_TUNING
    for (int variant : {0, 1, 2, 5}) {
#else
    for (int variant : {0, 12}) {
#endif
        timeit(Mtok, 2048, 2048, 0, 1, variant, 1, 0, 6);
        timeit(Mtok, 4096, 2048, 0, 1, variant, 1, LSTC_EPI_BIAS | LSTC_EPI_RELU, 5);
        timeit(Mtok, 2048, 4096, 0, 1, variant, 1, LSTC_EPI_BIAS | LSTC_EPI_DROPOUT | LSTC_EPI_RESIDUAL, 5);
        timeit(Mtok, 2048, 2048, 0, 0, variant, 1, 0, 6);
        
        timeit(2048, 2048, Mtok, 1, 0, variant, 4, 0, 6);
        timeit(4096, 2048, Mtok, 1, 0, variant, 2, 0, 5);
    }
    return fails;
}
