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

// Seeds are LINEAR in the host's call counter (functional.next_seed: a captured step replays with seed + device word), so the
// avalanche happens here: splitmix64's finaliser over the whole 64-bit seed, k0 / k1 = its two halves.  Consecutive seeds - the
// dropout sites of one step, the steps of one run - get unrelated (k0, k1) pairs, hence independent masks (with k1 taken from the
// seed's high word alone it was constant over a run and every mask was an XOR-shifted window of one table: ADVICE r3).
// Wave-uniform scalar work, once per kernel (drop_key_now) or on the host (make_drop_key).
__host__ __device__ inline void drop_key_mix(DropKey& k, uint64_t seed) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k.k0 = (uint32_t)z;
    k.k1 = (uint32_t)(z >> 32);
}

// lstc_dropout_seed_device (api.hip): while set, every launch of this process that draws a dropout mask carries the pointer
// and re-derives its key from seed + *pointer ON THE DEVICE - a captured (hipGraph) step replays with fresh masks.
// (library-internal cross-unit symbols are hidden from the dynamic symbol table: the C ABI is include/lstc_hip.h)
__attribute__((visibility("hidden"))) const uint64_t* lstc_seed_dev_current();

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

// ---------------------------------------------------------------------------------- float4 GEMM epilogue (32x32 f32 accumulators)
// Shared by gemm_f32.hip and gemm_pk.hip.  A v_mfma_*_32x32 accumulator register r of a lane holds row (r & 3) + 8 (r >> 2) + 4 h,
// column l31: stored as it lies, every lane writes 4 bytes per store and a wave-level store touches two 128-B segments (64 store
// instructions per 128x128 tile and lane; 12 us of a 235-us tile on the exact-f32 kernel).  Transposing each 4-register group
// across its lane quad (two DPP quad_perm steps) gives a lane four consecutive columns of ONE row: 16 global_store_dwordx4
// per lane, every wave-level store = 8 rows x 128-B full lines, bias / residual / ReLU-mask operands as float4 loads.
#ifdef __HIPCC__
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
// 4 x 4 transpose across (4 registers) x (the 4 lanes of a quad): afterwards register k of lane c holds what register c of lane k held
__device__ __forceinline__ void quad_transpose(float& r0, float& r1, float& r2, float& r3, int c) {
    { const float x = (c & 1) ? r0 : r1; const float y = quad_dpp<0xB1>(x); if (c & 1) r0 = y; else r1 = y; }    // quad_perm [1,0,3,2]
    { const float x = (c & 1) ? r2 : r3; const float y = quad_dpp<0xB1>(x); if (c & 1) r2 = y; else r3 = y; }
    { const float x = (c & 2) ? r0 : r2; const float y = quad_dpp<0x4E>(x); if (c & 2) r0 = y; else r2 = y; }    // quad_perm [2,3,0,1]
    { const float x = (c & 2) ? r1 : r3; const float y = quad_dpp<0x4E>(x); if (c & 2) r1 = y; else r3 = y; }
}

// Epilogue of one wave's (32 TM) x (32 TN) block whose top-left element is (mw0, nw0): quad-transposed float4 form (see
// gemm_f32_persist_kernel).  Needs N, ldc (and the operand's ld) multiples of 4 and 16-B aligned pointers; AUX = the launch has
// exactly one per-element operand (residual | ReLU-mask source | accumulate target).
struct EpiArgs {       // passed BY VALUE: a reference to the kernel's (modified) parameter copy would pin that struct in scratch memory
    float* C;
    const float* bias;
    const float* aux;  // the one per-element operand: residual | ReLU-mask source | accumulate target (AUX)
    int M, N, ldc, ldaux, flags, row_off;
    float alpha;
    DropKey dk;
};
__device__ __forceinline__ EpiArgs make_epi_args(float* C, const float* bias, const float* res, const float* relu_src,
                                                  int M, int N, int ldc, int ldr, int ld_relu, int flags, int row_off, float alpha, DropKey dk) {
    EpiArgs e;
    e.C = C; e.bias = bias;
    e.aux = (flags & LSTC_EPI_RESIDUAL) ? res : (flags & LSTC_EPI_RELU_MASK) ? relu_src : C;
    e.ldaux = (flags & LSTC_EPI_RESIDUAL) ? ldr : (flags & LSTC_EPI_RELU_MASK) ? ld_relu : ldc;
    e.M = M; e.N = N; e.ldc = ldc; e.flags = flags; e.row_off = row_off; e.alpha = alpha; e.dk = dk;
    return e;
}
#define LSTC_EPI_ARGS(p) make_epi_args((p).C, (p).bias, (p).res, (p).relu_src, (p).M, (p).N, (p).ldc, (p).ldr, (p).ld_relu, (p).flags, (p).row_off, (p).alpha, (p).dk)

template <int TM, int TN, bool AUX>
__device__ __forceinline__ void epilogue_f4(const EpiArgs p, floatx16 (&acc)[TM][TN], int mw0, int nw0, int lane) {
    const int l31 = lane & 31, h = lane >> 5;
    // ---- epilogue: quad-transposed, one float4 of a row per lane and register group.  Straight-line code: the per-element
    // operand (AUX: residual, ReLU-mask source or the accumulate target - at most one of them on this kernel) is loaded
    // unconditionally from clamped addresses, four groups ahead of its use, and only the store is predicated - a load inside
    // a flag branch makes the compiler drain vmcnt(0) at every join (16 serialized store round trips per tile, and the next
    // tile's prefetch with them).
    const int flags = p.flags;
    const int c = lane & 3, q = l31 >> 2;
    const float* aux = p.aux;
    const int ldaux = p.ldaux;
    const bool f_relu = flags & LSTC_EPI_RELU, f_drop = flags & LSTC_EPI_DROPOUT, f_mask = flags & LSTC_EPI_RELU_MASK;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = nw0 + j * 32 + 4 * q;
        const int colc = min(col, p.N - 4);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (flags & LSTC_EPI_BIAS) bv = *reinterpret_cast<const float4*>(p.bias + colc);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = mw0 + i * 32 + 4 * h + c;
            float4 ax[4];
            if constexpr (AUX) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    ax[g] = *reinterpret_cast<const float4*>(aux + (size_t)min(rbase + 8 * g, p.M - 1) * ldaux + colc);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float r0 = acc[i][j][4 * g + 0], r1 = acc[i][j][4 * g + 1], r2 = acc[i][j][4 * g + 2], r3 = acc[i][j][4 * g + 3];
                quad_transpose(r0, r1, r2, r3, c);
                const int row = rbase + 8 * g;
                float4 v = make_float4(r0 * p.alpha + bv.x, r1 * p.alpha + bv.y, r2 * p.alpha + bv.z, r3 * p.alpha + bv.w);
                if (f_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (f_drop) {
                    const uint32_t idx = (uint32_t)(row + p.row_off) * (uint32_t)p.N + (uint32_t)col;
                    v.x = drop_keep(idx, p.dk) ? v.x * p.dk.scale : 0.f;
                    v.y = drop_keep(idx + 1, p.dk) ? v.y * p.dk.scale : 0.f;
                    v.z = drop_keep(idx + 2, p.dk) ? v.z * p.dk.scale : 0.f;
                    v.w = drop_keep(idx + 3, p.dk) ? v.w * p.dk.scale : 0.f;
                }
                if constexpr (AUX) {
                    const float4 x = ax[g];
                    if (f_mask) {
                        v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f; v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
                    } else {                                 // residual or accumulate: both add the operand
                        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                    }
                }
                if (row < p.M && col < p.N) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = v;
            }
        }
    }
}
#endif
