// Shared device/host helpers for liblstc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/lstc_hip.h"

#define LSTC_WAVE 64

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- dropout RNG
// Counter-based keep/drop decision for flat element index i under a 64-bit seed.  Two-multiply
// avalanche hash (cheap enough to sit in a GEMM epilogue); keep iff hash >= p * 2^32.
struct DropKey {
    uint32_t k0, k1, thr;
    float scale;
    uint64_t seed;                 // the seed k0 / k1 were derived from
    const uint64_t* seed_dev;      // device word ADDED to the seed when the kernel runs (lstc_dropout_seed_device), or NULL
};

__host__ __device__ inline void drop_key_mix(DropKey& k, uint64_t seed) {
    k.k0 = (uint32_t)(seed & 0xffffffffu) * 0x9E3779B1u + 0x7F4A7C15u;
    k.k1 = (uint32_t)(seed >> 32) * 0x85EBCA77u + 0x165667B1u;
}

// lstc_dropout_seed_device (api.hip): while set, every launch of this process that draws a dropout mask carries the pointer
// and re-derives its key from seed + *pointer ON THE DEVICE - a captured (hipGraph) step replays with fresh masks.
const uint64_t* lstc_seed_dev_current();

inline DropKey make_drop_key(float p, uint64_t seed) {
    DropKey k;
    drop_key_mix(k, seed);
    double t = (double)p * 4294967296.0;
    k.thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    k.scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
    k.seed = seed;
    k.seed_dev = lstc_seed_dev_current();
    return k;
}

// Called ONCE at the top of a kernel (wave-uniform scalar work): the key this launch uses.
__device__ inline DropKey drop_key_now(DropKey k) {
    if (k.seed_dev) drop_key_mix(k, k.seed + *k.seed_dev);
    return k;
}

__host__ __device__ inline uint32_t drop_hash(uint32_t i, const DropKey& k) {
    uint32_t h = i ^ k.k0;
    h *= 0x9E3779B1u;
    h ^= h >> 15;
    h += k.k1;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    h *= 0xC2B2AE3Du;
    h ^= h >> 16;
    return h;
}

__host__ __device__ inline bool drop_keep(uint32_t i, const DropKey& k) { return drop_hash(i, k) >= k.thr; }

// ---------------------------------------------------------------------------------- reductions
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x == NT (multiple of 64); `red` is NT/64 floats of LDS.
template <int NT>
__device__ inline float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// ---------------------------------------------------------------------------------- packed bf16 operands (lstc_pack1)
// Element offset of (row, k) inside an lstc_pack1 buffer whose row blocks hold KBp 32-k tiles: 128-row x 32-k tiles of 4096
// elements, 64-B rows, 16-B chunk index XOR (row >> 2) & 3 (csrc/gemm_bf16p.hip).  k must be a multiple of 4 for the 8-B
// (4-element) stores of the row-wise producers.
__host__ __device__ static inline size_t p1_offset(int64_t row, int k, int KBp) {
    const int rr = (int)(row & 127), kb = k >> 5, ch = (k & 31) >> 3;
    return ((size_t)(row >> 7) * KBp + kb) * 4096 + (size_t)rr * 32 + (size_t)((ch ^ ((rr >> 2) & 3)) << 3) + (k & 7);
}

static inline int lstc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// One-time per-DEVICE launch setup.  Kernel attributes (the dynamic-LDS opt-in above 64 KB) belong to a device, so a process
// that launches on a second GPU must set them again there:
//     static LstcDevOnce once;  const int dev = once.begin();  if (dev >= 0) { hipFuncSetAttribute(...); once.end(dev); }
// The guarded calls are idempotent: two host threads racing through their first launch merely repeat them; the flag word is atomic.
struct LstcDevOnce {
    std::atomic<uint64_t> mask{0};
    int begin() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        return ((mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull) ? -1 : dev;
    }
    void end(int dev) { mask.fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

__host__ __device__ static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
