// Probe: how fast can a ROW-WISE kernel stream an lstc_pack1 operand [rows, d] (128-row x 32-k tiles, 64-B rows)?
//   A: one wave per row, lane = 16-B chunk c8 of the row: every 4 lanes read one tile row (64 B), 16 tiles per instruction
//   B: one wave per FOUR rows: lane = (tile l >> 4, row (l >> 2) & 3, chunk l & 3): 4 tiles x 256 contiguous bytes per instruction
//   C: one wave per EIGHT rows: lane = (tile l >> 5, row (l >> 2) & 7, chunk l & 3): 2 tiles x 512 contiguous bytes per instruction
// Each reads the pack, adds 1 to every bf16 pair (as u32) and writes a second pack the same way.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ size_t p1_off(int64_t row, int k, int KBp) {
    const int rr = (int)(row & 127), kb = k >> 5, ch = (k & 31) >> 3;
    return ((size_t)(row >> 7) * KBp + kb) * 4096 + (size_t)rr * 32 + (size_t)((ch ^ ((rr >> 2) & 3)) << 3);
}
template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t rows, int d) {
    const int lane = threadIdx.x & 63, KBp = d / 32;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    constexpr int RPW = MODE == 0 ? 1 : MODE == 1 ? 4 : 8;       // rows per wave iteration
    const int nchunk = d / 8;                                    // 16-B chunks per row
    const int per_lane = nchunk * RPW / 64;
    for (int64_t g = wave0; g * RPW < rows; g += nw) {
        uint4v v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i >= per_lane) break;
            int64_t r; int c8;
            if (MODE == 0) { r = g; c8 = lane + 64 * i; }
            else if (MODE == 1) { r = g * 4 + ((lane >> 2) & 3); c8 = ((lane >> 4) + 4 * i) * 4 + (lane & 3); }
            else { r = g * 8 + ((lane >> 2) & 7); c8 = ((lane >> 5) + 2 * i) * 4 + (lane & 3); }
            v[i] = *reinterpret_cast<const uint4v*>(x + p1_off(r, 8 * c8, KBp));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (i >= per_lane) break;
            int64_t r; int c8;
            if (MODE == 0) { r = g; c8 = lane + 64 * i; }
            else if (MODE == 1) { r = g * 4 + ((lane >> 2) & 3); c8 = ((lane >> 4) + 4 * i) * 4 + (lane & 3); }
            else { r = g * 8 + ((lane >> 2) & 7); c8 = ((lane >> 5) + 2 * i) * 4 + (lane & 3); }
            uint4v o = v[i]; o[0] += 1; o[1] += 1; o[2] += 1; o[3] += 1;
            *reinterpret_cast<uint4v*>(y + p1_off(r, 8 * c8, KBp)) = o;
        }
    }
}
int main() {
    const int64_t rows = 100352; const int d = 2048;
    const size_t n = (size_t)rows * d;
    uint16_t *x, *y, *big;
    hipMalloc(&x, n * 2 + 65536); hipMalloc(&y, n * 2 + 65536); hipMalloc(&big, (size_t)1 << 30);
    hipMemset(x, 1, n * 2); hipMemset(y, 0, n * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 2048, 4096}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipMemsetAsync(big, it, (size_t)1 << 30, 0);        // flush the Infinity Cache between runs
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, x, y, rows, d);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, x, y, rows, d);
                else hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, x, y, rows, d);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it > 0 && ms < best) best = ms;
            }
            printf("grid %4d  rows/wave %d: %.1f us  %.2f TB/s (read + write)\n", grid, mode == 0 ? 1 : mode == 1 ? 4 : 8, best * 1e3, 2.0 * n * 2 / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
