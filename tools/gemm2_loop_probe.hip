// Round 6: MAIN-LOOP-ONLY timing prototype of the "second-generation" bf16 GEMM form that DESIGN 8 has named since
// round 2 - 256 x 256 output tile per workgroup, FOUR waves, one per SIMD, each owning a 128 x 128 sub-tile (256 accumulator
// registers per lane: AGPR territory, 512 registers per lane in all) - against the shipped 8-wave form (2 waves per SIMD, 128 x 64 each).
// Same operand traffic as the shipped kernel: per K step of 64 the workgroup streams 64 KB (A 256 x 64 + B 256 x 64, bf16) into one of
// two LDS buffers with global_load_lds_dwordx4 (64 pieces of 1 KB: SIXTEEN per wave here, eight in the shipped kernel) and every wave
// reads its fragments with ds_read_b128 (32 per K step: 8 row tiles + 8 column tiles x 2 k halves; the shipped wave reads 24).
// What it answers: with ONE wave per SIMD nothing overlaps that wave's LDS-DMA issue and fragment reads with its MFMAs except the
// instruction stream itself - is the loop still faster than the shipped 1.33 us per K step (K = 2048 items, tools/r06_bf16p_stamps.sh)?
// No epilogue, no tile walk, products are not checked (operands are random bf16 in the 64-B-row tile layout; DMA sources stay L2 /
// Infinity-Cache resident like the weight operand of the real kernel - variant "hot" - or stream through a 1-GB buffer - "stream").
//   hipcc -O3 --offload-arch=gfx950 tools/gemm2_loop_probe.hip -o tools/gemm2_loop_probe && tools/gemm2_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BUF = 65536;                   // bytes of one K step's operands in LDS (A 32 KB + B 32 KB)
constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }

// DMA: 0 = none (MFMA + fragment reads only), 1 = every wave issues its 16 pieces per K step, spread 1 per 8 MFMAs,
//      2 = the same pieces issued as one burst at the top of the K step.   RD: fragment reads on / off.
template <int DMA, bool RD>
__global__ void __launch_bounds__(256, 1) loop4(const char* __restrict__ src, size_t src_bytes, float* __restrict__ out, int ksteps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1;                       // 2 x 2 waves of 128 x 128
    floatx4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses: 16-row x 32-k sub-tiles of 1 KB (64-B rows, 16-B chunk XOR (row >> 2) & 3), as csrc/gemm_bf16p.hip lays them out
    const int l15 = lane & 15, c16 = lane >> 4;
    const int frag_off = l15 * 64 + ((((0x9C >> (2 * c16)) & 3) ^ ((l15 >> 2) & 3)) * 16);
    const uint32_t lane_off = (uint32_t)lane * 16u;
    // this workgroup's DMA source: a window of the buffer (hot: 2 MB per workgroup re-read every 32 K steps; stream: walks the whole buffer)
    const size_t win = src_bytes / gridDim.x;
    const char* base = src + (size_t)blockIdx.x * win;
    bf16x8 fa[2][8], fb[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[k][i] = bf16x8{}; fb[k][i] = bf16x8{}; }
    auto dma_piece = [&](int ks, int piece, int buf) {              // piece 0..15 of this wave: 1 KB each
        const size_t goff = ((size_t)ks * BUF + (size_t)(wave * 16 + piece) * 1024) % win;
        const char* g_ = base + goff;
        const uint32_t l_ = (uint32_t)(buf * BUF + (wave * 16 + piece) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(g_), "s"(l_) : "memory");
    };
    if (DMA) {
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) dma_piece(0, pc, 0);
    }
    for (int ks = 0; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        if (DMA) __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));          // K step ks has landed (issued one step ago)
        __builtin_amdgcn_s_barrier();                               // ... for every wave; buffer cur ^ 1 is free (everybody finished step ks - 1)
        if (DMA == 2) {
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) dma_piece(ks + 1, pc, cur ^ 1);
        }
        const char* A = lds + cur * BUF + wr * 16384;               // the wave's 128 rows: 8 sub-tiles x 2 k halves
        const char* B = lds + cur * BUF + 32768 + wc * 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (RD) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    fa[kk][i] = *reinterpret_cast<const bf16x8*>(A + (i * 2 + kk) * 1024 + frag_off);
                    fb[kk][i] = *reinterpret_cast<const bf16x8*>(B + (i * 2 + kk) * 1024 + frag_off);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][rt], fb[kk][ct], acc[rt][ct], 0, 0, 0);
                if (DMA == 1) dma_piece(ks + 1, kk * 8 + rt, cur ^ 1);          // one piece per 8 MFMAs
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 256 + t] = s;
}

template <int DMA, bool RD>
void run(const char* name, const char* src, size_t bytes, float* out) {
    const int ksteps = 32 * 13, grid = 256;                         // = the K steps one persistent workgroup walks in a 100352 x 2048 x 2048 launch
    const size_t ldsb = 2 * BUF;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loop4<DMA, RD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 6; ++w) hipLaunchKernelGGL((loop4<DMA, RD>), grid, 256, ldsb, 0, src, bytes, out, ksteps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((loop4<DMA, RD>), grid, 256, ldsb, 0, src, bytes, out, ksteps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double flop = (double)grid * ksteps * 2.0 * 256 * 256 * 64;
    printf("%-78s %8.3f ms  %6.3f us per K step  %7.1f TFLOP/s-equivalent\n", name, ms, ms * 1e3 / ksteps, flop / ms / 1e9);
}

int main() {
    const size_t hot = (size_t)256 * 2 * 1024 * 1024, big = (size_t)1 << 30;
    char* src; float* out;
    CK(hipMalloc(&src, big)); CK(hipMalloc(&out, 256 * 256 * 4));
    {   // random bf16 bit patterns with sane exponents (DVFS: never time MFMA on zeros)
        std::vector<uint16_t> h(1 << 22);
        uint32_t x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00u + ((x >> 16) & 0x03ffu) + ((x >> 31) << 15)); }
        for (size_t o = 0; o < big; o += h.size() * 2) CK(hipMemcpy(src + o, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    printf("one wave per SIMD, 128 x 128 per wave, 256 workgroups x %d K steps of 64 (the shipped 8-wave loop: 1.33 us per K step with LDS-DMA, 1.17 without)\n", 32 * 13);
    run<0, false>("MFMA only (fragments stay in registers)", src, hot, out);
    run<0, true>("MFMA + 32 ds_read_b128 per wave and K step, no DMA", src, hot, out);
    run<1, true>("+ 16 LDS-DMA pieces per wave, one per 8 MFMAs, sources cache-resident (2 MB / workgroup)", src, hot, out);
    run<2, true>("+ 16 LDS-DMA pieces per wave as one burst per K step, sources cache-resident", src, hot, out);
    run<1, true>("+ 16 LDS-DMA pieces per wave, one per 8 MFMAs, sources streamed from a 1-GB buffer", src, big, out);
    run<2, true>("+ 16 LDS-DMA pieces per wave as one burst per K step, sources streamed", src, big, out);
    return 0;
}
