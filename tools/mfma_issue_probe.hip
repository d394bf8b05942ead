// Micro-benchmark: what each kind of companion instruction costs the f32 MFMA pipe (gfx950).
// Each wave loops over groups of 4 independent v_mfma_f32_32x32x2_f32 (or 8 v_mfma_f32_16x16x4_f32) and, per group,
// issues R ds_read_b128, W ds_write_b128 and G global_load_dwordx4.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int R, int W, int G, bool SMALL>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ g, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    floatx16 acc[4];
    floatx4 acs[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acs[i][r] = 0.f;
    float a = t * 0.001f, b = 1.f - t * 0.002f;
    float4 rd[R > 0 ? R : 1], st[W > 0 ? W : 1], gl[G > 0 ? G : 1];
    for (int i = 0; i < (W > 0 ? W : 1); ++i) st[i] = make_float4(a, b, a, b);
    const float4* gp = reinterpret_cast<const float4*>(g) + (size_t)blockIdx.x * 4096 + t;
    float sink = 0.f;
    // software-pipelined: the loads issued in iteration it are consumed in iteration it+1 (register double buffer),
    // so what is measured is the issue / write-back cost of the companion instructions, not their latency
    float4 rd2[R > 0 ? R : 1], gl2[G > 0 ? G : 1];
    for (int i = 0; i < (R > 0 ? R : 1); ++i) rd2[i] = make_float4(0, 0, 0, 0);
    for (int i = 0; i < (G > 0 ? G : 1); ++i) gl2[i] = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float4* rcur = ph ? rd2 : rd; float4* rprev = ph ? rd : rd2;
            float4* gcur = ph ? gl2 : gl; float4* gprev = ph ? gl : gl2;
#pragma unroll
            for (int i = 0; i < R; ++i) rcur[i] = *reinterpret_cast<const float4*>(lds + ((t * 4 + i * 1024 + (it + ph) * 16) & 8191));
#pragma unroll
            for (int i = 0; i < W; ++i) *reinterpret_cast<float4*>(lds + 8192 + ((t * 4 + i * 1024) & 8191)) = st[i];
#pragma unroll
            for (int i = 0; i < G; ++i) gcur[i] = gp[(size_t)(((it + ph) * G + i) & 15) * 256];
            if (SMALL) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acs[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < R; ++i) sink += rprev[i].x;
#pragma unroll
            for (int i = 0; i < G; ++i) sink += gprev[i].y;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = sink;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 16; ++i) s += acs[i][0];
    out[blockIdx.x * 256 + t] = s;
}

template <int R, int W, int G, bool SMALL>
void run(const char* name, int wgs_per_cu, const float* g, float* out) {
    const int iters = 20000, grid = 256 * wgs_per_cu;
    const size_t ldsb = wgs_per_cu == 1 ? 96 * 1024 : 65536;       // LDS size pins residency: 1 or 2 workgroups per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<R, W, G, SMALL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<R, W, G, SMALL>), grid, 256, ldsb, 0, g, out, iters);
    CK(hipEventRecord(e0)); 
    hipLaunchKernelGGL((probe<R, W, G, SMALL>), grid, 256, ldsb, 0, g, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)grid * 4 /*waves*/ * iters * 8 * 4096.0;     // 8 x (32x32x2) or 16 x (16x16x4) = 32768 FLOP per wave-iter
    printf("%-46s %d WG/CU: %7.2f TFLOP/s (%.1f%% of 157.3)\n", name, wgs_per_cu, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
}

int main() {
    float *g, *out; CK(hipMalloc(&g, 256 * 2 * 4096 * 16)); CK(hipMalloc(&out, 512 * 256 * 4));
    CK(hipMemset(g, 0, 256 * 2 * 4096 * 16));
    for (int wg = 1; wg <= 2; ++wg) {
        run<0, 0, 0, false>("32x32x2 only", wg, g, out);
        run<0, 0, 0, true>("16x16x4 only", wg, g, out);
        run<2, 0, 0, false>("32x32x2 + 2 ds_read_b128 / 8 MFMA", wg, g, out);
        run<4, 0, 0, false>("32x32x2 + 4 ds_read_b128 / 8 MFMA", wg, g, out);
        run<0, 1, 0, false>("32x32x2 + 1 ds_write_b128 / 8 MFMA", wg, g, out);
        run<0, 2, 0, false>("32x32x2 + 2 ds_write_b128 / 8 MFMA", wg, g, out);
        run<0, 0, 1, false>("32x32x2 + 1 global_load_x4 / 8 MFMA", wg, g, out);
        run<0, 0, 2, false>("32x32x2 + 2 global_load_x4 / 8 MFMA", wg, g, out);
        run<2, 1, 1, false>("32x32x2 + 2 rd + 1 wr + 1 gl / 8 MFMA (GEMM mix)", wg, g, out);
        run<2, 1, 1, true>("16x16x4 + 2 rd + 1 wr + 1 gl / 16 MFMA (GEMM mix)", wg, g, out);
    }
    return 0;
}
