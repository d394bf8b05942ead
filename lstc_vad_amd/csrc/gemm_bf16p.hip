// bf16 GEMM over PACKED bf16 operands (LstcGemmDesc.dtype = LSTC_BF16P): the "bf16" compute mode of BASELINE.json
// configs 3 / 5 for every large product (fp32 master weights / activations / accumulation, bf16 matrix cores).
//
// gemm_bf16c.hip keeps the operands f32 in HBM and rounds them while staging: it moves 4 B per element through L2 -> LDS and
// is bound there (425 TFLOP/s of 2500).  Here an operand is rounded ONCE (lstc_pack1, RNE) into 128-row x 32-k tiles that are
// byte for byte the LDS image the kernel reads (64-B rows, 16-B chunk index XOR (row >> 2) & 3: the 16 lanes of every
// ds_read_b128 lane group land on 16 distinct 16-B slots), so the GEMM streams 2 B per element with global_load_lds_dwordx4
// (1 KB contiguous per wave-instruction, no staging registers, no ds_write) and the same packs serve three products:
//   forward / input gradient   C = A B^T      packs of [M, K] and [N, K]                  (NT form, fragments by ds_read_b128)
//   weight gradient            C = A^T B      packs of the SOURCES [K = tokens, M], [K, N] (TR form: the contraction runs
//                                              along the packs' rows; fragments by ds_read_b64_tr_b16, the transpose read)
//
// Schedule (cdna_hip_programming.md 5, "256^2 8-phase template", rebuilt on pre-tiled operands): 256x256 output tile, K step
// 64, 8 waves = 2 (M) x 4 (N), 128x64 per wave, two 64-KB LDS buffers, one workgroup per CU.  A K step is four PHASES, one
// per 64x32 quadrant of the wave's tile; a phase = LOAD segment {ds_read the quadrant's fragments, issue 2 LDS-DMA pieces}
// - barrier - COMPUTE segment {8 x v_mfma_f32_32x32x16_bf16} - barrier.  Waves 4-7 run one barrier behind waves 0-3, so on
// every SIMD one wave's COMPUTE overlaps its partner's LOAD and the matrix pipe alternates between the two.  The LDS-DMA
// stream runs 5-6 phases ahead of its use with ONE counted s_waitcnt vmcnt(4) per K step (never 0 in the loop); raw
// s_barrier (a __syncthreads() fence would drain every DMA in flight).
//
// LDS buffer = 16 slots of 4 KB (64 rows x 64 B): slots 0-7 operand A, 8-15 operand B.
//   NT: slot = 2 * r64 + kt2  (rows 64 r64 .. +63 of the operand's 256 rows, k tile kt2 of the K step)
//   TR: slot = feature block fb (64 tokens x 32 features)
// Staging units (16 pieces of 1 KB each, 2 per wave), in the order their LDS regions fall free:
//   U0 = A rows/features of every wave's FIRST 64 (read in phase 0), U1 = B first 32 (phase 0), U2 = B second 32 (phase 1),
//   U3 = A second 64 (phase 2).  Issue at (tile t, phase p): p0 U2(t+1), p1 U3(t+1), p2 U0(t+2), p3 U1(t+2) - each region was
//   last read >= 2 phases earlier (WAR), each unit lands >= 5 phases before its first read (RAW: vmcnt(4) at p3 + barrier).
#include <type_traits>
#include "lstc_common.h"

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NT8 = 512;                 // 8 waves
constexpr int P1_TILE = 4096;            // bf16 elements of one packed tile (128 rows x 32 k = 8 KB)
constexpr int P1_SLOT = 2048;            // 64 rows x 32 k (4 KB)
constexpr int P1_BUF = 16 * P1_SLOT;     // 64 KB
constexpr int P1_SLACK = 65536;          // bytes after the tiles (TR feature-block over-reads of ragged outputs)

// ---- pack, K-contiguous source [rows, K] (ld): one workgroup per row block and 4 consecutive k tiles.
constexpr int P1_KPB = 4;
__global__ void __launch_bounds__(256) pack1_kc_kernel(const float* __restrict__ x, int rows, int K, long long ld,
                                                       bf16_t* __restrict__ out, int KBp) {
    const int kgroups = (KBp + P1_KPB - 1) / P1_KPB;
    const int kb0 = (blockIdx.x % kgroups) * P1_KPB, rb = blockIdx.x / kgroups;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    float4 va[P1_KPB][2], vb[P1_KPB][2];
#pragma unroll
    for (int tI = 0; tI < P1_KPB; ++tI)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s = threadIdx.x + q * 256;     // chunk id inside the tile: row r = s / 4, chunk c = s % 4
            const int row = rb * 128 + (s >> 2), k0 = (kb0 + tI) * 32 + (s & 3) * 8;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a;
            if (row < rows && k0 < K) {
                const float* px = x + (size_t)row * ld + k0;
                if (k0 + 8 <= K && vec) {
                    a = *reinterpret_cast<const float4*>(px);
                    b4 = *reinterpret_cast<const float4*>(px + 4);
                } else {
                    float t8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t8[j] = (k0 + j < K) ? px[j] : 0.f;
                    a = make_float4(t8[0], t8[1], t8[2], t8[3]); b4 = make_float4(t8[4], t8[5], t8[6], t8[7]);
                }
            }
            va[tI][q] = a; vb[tI][q] = b4;
        }
#pragma unroll
    for (int tI = 0; tI < P1_KPB; ++tI) {
        if (kb0 + tI >= KBp) break;
        bf16x8* o = reinterpret_cast<bf16x8*>(out) + ((size_t)rb * KBp + kb0 + tI) * 512;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s = threadIdx.x + q * 256, r = s >> 2, c = s & 3;
            const float v[8] = {va[tI][q].x, va[tI][q].y, va[tI][q].z, va[tI][q].w, vb[tI][q].x, vb[tI][q].y, vb[tI][q].z, vb[tI][q].w};
            bf16x8 hh;
#pragma unroll
            for (int j = 0; j < 8; ++j) hh[j] = (bf16_t)v[j];
            o[r * 4 + (c ^ ((r >> 2) & 3))] = hh;
        }
    }
}

// ---- pack, k-major source [K, rows] (ld): the packed operand's row index runs along the source's contiguous dimension.
__global__ void __launch_bounds__(256) pack1_km_kernel(const float* __restrict__ x, int rows, int K, long long ld,
                                                       bf16_t* __restrict__ out, int KBp) {
    const int kb = blockIdx.x % KBp, rb = blockIdx.x / KBp;
    bf16x8* o = reinterpret_cast<bf16x8*>(out) + (size_t)blockIdx.x * 512;
    const int r = threadIdx.x & 127, half = threadIdx.x >> 7;
    const int row = rb * 128 + r;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int c = half * 2 + cc, k0 = kb * 32 + c * 8;
        bf16x8 hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) hh[j] = (bf16_t)((row < rows && k0 + j < K) ? x[(size_t)(k0 + j) * ld + row] : 0.f);
        o[r * 4 + (c ^ ((r >> 2) & 3))] = hh;
    }
}

struct P1Params {
    const bf16_t* A;
    const bf16_t* B;
    float* C;
    const float* bias;
    const float* res;
    const float* relu_src;
    int M, N, ldc, ldr, ld_relu, flags;
    float alpha;
    DropKey dk;
    int tilesN, nsteps, steps_per_split;     // K steps of 64
    int KBa, KBb;                            // 32-k tiles per 128-row block of the A / B pack (even)
    long long split_stride;                  // elements between the outputs of consecutive K splits (0: atomics into one C)
    int debug;                               // -DLSTC_TUNING builds only: 1 = skip the epilogue (timing ablation)
};

template <bool TR>
__global__ void __launch_bounds__(NT8, 2) gemm_bf16p_kernel(const P1Params p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_p1[];
    bf16_t* const smem = smem_p1;
    int pid = blockIdx.x;
    {   // XCD-aware bijective remap: each XCD works on a contiguous run of tiles (N fastest) and keeps their A panel in its L2
        const int nwg = gridDim.x, xcd = pid & 7, idx = pid >> 3, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mb = pid / p.tilesN, nb = pid % p.tilesN;
    const int kt0 = blockIdx.y * p.steps_per_split;
    const int nkt = min(p.nsteps, kt0 + p.steps_per_split) - kt0;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- LDS-DMA: per unit one wave-uniform global base (SGPRs) + LDS byte address; pieces j = 0, 1 are consecutive KBs
    // (the instruction's immediate offset applies to both sides).  Element offsets.
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const int wa = wave >> 2, wb = (wave >> 1) & 1, wj = wave & 1;
    // slot and first piece of this wave inside unit u (0: A first, 1: B first, 2: B second, 3: A second)
    auto unit_slot = [&](int u) -> int {
        if (u == 0 || u == 3) return 4 * wa + (u == 3 ? 2 : 0) + wb;
        return TR ? 8 + 2 * (wave >> 1) + (u == 2 ? 1 : 0) : 8 + wave;
    };
    auto unit_piece = [&](int u) -> int {
        if (u == 0 || u == 3) return 2 * wj;
        return TR ? 2 * wj : (u == 2 ? 2 : 0);
    };
    // global element address of the slot's first row for K step kt (absolute step index)
    auto slot_gaddr = [&](int slot, int kt) -> const bf16_t* {
        const bool isb = slot >= 8;
        const int s = slot & 7;
        const bf16_t* base = isb ? p.B : p.A;
        const int KBx = isb ? p.KBb : p.KBa, ob = isb ? nb : mb;
        if (TR) return base + ((size_t)(kt >> 1) * KBx + 8 * ob + s) * P1_TILE + (kt & 1) * P1_SLOT;
        return base + ((size_t)(2 * ob + (s >> 2)) * KBx + 2 * kt + (s & 1)) * P1_TILE + ((s >> 1) & 1) * P1_SLOT;
    };
#define P1_DMA_UNIT(u, ktl, buf)                                                                                       \
    do {                                                                                                               \
        const int slot_ = unit_slot(u), pc_ = unit_piece(u);                                                           \
        const bf16_t* g_ = slot_gaddr(slot_, kt0 + (ktl)) + pc_ * 512;                                                 \
        const uint32_t l_ = (uint32_t)(((buf) * P1_BUF + slot_ * P1_SLOT + pc_ * 512) * 2);                            \
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" \
                     :: "v"(lane_off), "s"(g_), "s"(l_) : "memory");                                                   \
    } while (0)

    // ---- fragment reads.  NT: row (i & 1) * 32 + l31 of slot 2 * (2 wr + (i >> 1)) + (q >> 1), 16-B chunk (2 h + (q & 1)) ^ swz.
    // TR: slot = feature block; tokens 16 q + 8 h + trq (+4), transposed read (lane receives feature lane & 31).
    const int swz = (l31 >> 2) & 3;
    const int nt_off0 = l31 * 32 + ((2 * h) ^ swz) * 8, nt_off1 = l31 * 32 + ((2 * h + 1) ^ swz) * 8;
    const int trq = (lane >> 2) & 3, trchunk = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), trsub = (lane & 1) * 4;
    const int tr_off0 = (8 * h + trq) * 32 + ((trchunk ^ (2 * h)) * 8) + trsub;
    const int tr_off1 = (8 * h + trq + 4) * 32 + ((trchunk ^ (2 * h + 1)) * 8) + trsub;
    auto rd_tr = [&](const bf16_t* slot, int q) -> bf16x8 {
        typedef short short4v __attribute__((ext_vector_type(4)));
        typedef short short8v __attribute__((ext_vector_type(8)));
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(slot + q * 512 + tr_off0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(slot + q * 512 + tr_off1));
        short8v f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return __builtin_bit_cast(bf16x8, f);
    };
    auto rd_a = [&](const bf16_t* buf, int i, int q) -> bf16x8 {
        if (TR) return rd_tr(buf + (4 * wr + i) * P1_SLOT, q);
        const bf16_t* s = buf + (2 * (2 * wr + (i >> 1)) + (q >> 1)) * P1_SLOT + (i & 1) * 1024;
        return *reinterpret_cast<const bf16x8*>(s + ((q & 1) ? nt_off1 : nt_off0));
    };
    auto rd_b = [&](const bf16_t* buf, int j, int q) -> bf16x8 {
        if (TR) return rd_tr(buf + (8 + 2 * wc + j) * P1_SLOT, q);
        const bf16_t* s = buf + (8 + 2 * wc + (q >> 1)) * P1_SLOT + j * 1024;
        return *reinterpret_cast<const bf16x8*>(s + ((q & 1) ? nt_off1 : nt_off0));
    };
    bf16x8 fa[4][2], fb[2][4];            // A: [k16 step][row tile of the current half]; B: [column tile][k16 step]

    // ---- prologue: tile 0 complete + U0, U1 of tile 1 (the in-flight state every tile starts from)
    P1_DMA_UNIT(0, 0, 0); P1_DMA_UNIT(1, 0, 0); P1_DMA_UNIT(2, 0, 0); P1_DMA_UNIT(3, 0, 0);
    if (nkt > 1) {
        P1_DMA_UNIT(0, 1, 1); P1_DMA_UNIT(1, 1, 1);
        __builtin_amdgcn_s_waitcnt(0x0F74);          // vmcnt(4): this wave's 8 pieces of tile 0 have landed
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();       // waves 4-7 run one barrier behind: their LOAD beside the partner's COMPUTE

#define P1_MMA(ih, jj)                                                                                                 \
    do {                                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                   \
            acc[2 * (ih) + ii][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][ii], fb[jj][q], acc[2 * (ih) + ii][jj], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
    } while (0)
#define P1_SYNC_COMPUTE(ih, jj)                                                                                        \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);          /* lgkmcnt(0): this phase's fragments are in registers */         \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        P1_MMA(ih, jj);                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)

    auto tile = [&](int tl, auto bufc) {
        constexpr int CUR = decltype(bufc)::value;
        const bf16_t* cur = smem + CUR * P1_BUF;
        const bool has1 = tl + 1 < nkt, has2 = tl + 2 < nkt;
        // ---- phase 0: quadrant (rows first 64, cols first 32)
#pragma unroll
        for (int q = 0; q < 4; ++q) fb[0][q] = rd_b(cur, 0, q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { fa[q][0] = rd_a(cur, 0, q); fa[q][1] = rd_a(cur, 1, q); }
        if (has1) P1_DMA_UNIT(2, tl + 1, CUR ^ 1);
        P1_SYNC_COMPUTE(0, 0);
        // ---- phase 1: (first 64 rows, second 32 cols)
#pragma unroll
        for (int q = 0; q < 4; ++q) fb[1][q] = rd_b(cur, 1, q);
        if (has1) P1_DMA_UNIT(3, tl + 1, CUR ^ 1);
        P1_SYNC_COMPUTE(0, 1);
        // ---- phase 2: (second 64 rows, second 32 cols)
#pragma unroll
        for (int q = 0; q < 4; ++q) { fa[q][0] = rd_a(cur, 2, q); fa[q][1] = rd_a(cur, 3, q); }
        if (has2) P1_DMA_UNIT(0, tl + 2, CUR);
        P1_SYNC_COMPUTE(1, 1);
        // ---- phase 3: (second 64 rows, first 32 cols): no fragment reads; tile t+1 must have landed before the next read
        if (has2) {
            P1_DMA_UNIT(1, tl + 2, CUR);
            __builtin_amdgcn_s_waitcnt(0x0F74);      // vmcnt(4): everything but U0, U1 of tile t+2
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
        }
        P1_SYNC_COMPUTE(1, 0);
    };
    int tl = 0;
    for (; tl + 1 < nkt; tl += 2) {
        tile(tl, std::integral_constant<int, 0>{});
        tile(tl + 1, std::integral_constant<int, 1>{});
    }
    if (tl < nkt) tile(tl, std::integral_constant<int, 0>{});
    if (wr == 0) __builtin_amdgcn_s_barrier();       // balance the stagger
#undef P1_DMA_UNIT
#undef P1_MMA
#undef P1_SYNC_COMPUTE

    // ---- epilogue (semantics of gemm_f32.hip)
#ifdef LSTC_TUNING
    if (p.debug & 1) {       // timing ablation: keep the accumulators live, store nothing
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 1.2345e-30f) p.C[0] = s;
        return;
    }
#endif
    const int flags = p.flags;
    const bool atomic = gridDim.y > 1 && p.split_stride == 0;
    float* const Cz = p.C + (size_t)blockIdx.y * p.split_stride;
    const float alpha = p.alpha;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = nb * 256 + wc * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = (flags & LSTC_EPI_BIAS) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rbase = mb * 256 + wr * 128 + i * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row >= p.M) continue;
                float v = acc[i][j][r] * alpha;
                float* cp = Cz + (size_t)row * p.ldc + col;
                if (atomic) {
                    atomicAdd(cp, v);
                    continue;
                }
                v += bv;
                if (flags & LSTC_EPI_RELU) v = fmaxf(v, 0.f);
                if (flags & LSTC_EPI_DROPOUT) {
                    const uint32_t idx = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;
                    v = drop_keep(idx, p.dk) ? v * p.dk.scale : 0.f;
                }
                if (flags & LSTC_EPI_RESIDUAL) v += p.res[(size_t)row * p.ldr + col];
                if (flags & LSTC_EPI_RELU_MASK) v = p.relu_src[(size_t)row * p.ld_relu + col] > 0.f ? v : 0.f;
                if (flags & LSTC_EPI_ACCUM) v += *cp;
                *cp = v;
            }
        }
    }
}

inline int64_t p1_rbp(int64_t rows) { const int64_t rb = (rows + 127) / 128; return rb + (rb & 1); }
inline int64_t p1_kbp(int64_t K) { const int64_t kb = (K + 31) / 32; return kb + (kb & 1); }

}  // namespace

// Packed-operand bf16 GEMM behind lstc_gemm (dtype LSTC_BF16P): d->A / d->B point to lstc_pack1 outputs.
int lstc_gemm_bf16p_impl(const LstcGemmDesc* d, hipStream_t st) {
    if (!d->A || !d->B || !d->C) return LSTC_E_NULL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->ldc < d->N) return LSTC_E_SHAPE;
    if (d->batch > 1) return LSTC_E_UNSUPPORTED;
    if ((d->flags & LSTC_EPI_BIAS) && !d->bias) return LSTC_E_NULL;
    if ((d->flags & LSTC_EPI_RESIDUAL) && (!d->residual || d->ldr < d->N)) return LSTC_E_NULL;
    if ((d->flags & LSTC_EPI_RELU_MASK) && (!d->relu_src || d->ld_relu < d->N)) return LSTC_E_NULL;
    if ((d->flags & LSTC_EPI_DROPOUT) && (uint64_t)d->M * (uint64_t)d->N > 0xffffffffull) return LSTC_E_RANGE;
    if (!aligned16(d->A) || !aligned16(d->B)) return LSTC_E_ALIGN;
    const int splits = d->split_k > 1 ? d->split_k : 1;
    if (splits > 1 && d->flags != 0) return LSTC_E_UNSUPPORTED;
    // (transA, transB) = (0, 1): A, B are packs of [M, K], [N, K];  (1, 0): packs of the k-major sources [K, M], [K, N]
    const bool tr = d->transA != 0 && d->transB == 0;
    if (!tr && !(d->transA == 0 && d->transB != 0)) return LSTC_E_UNSUPPORTED;
    if (tr && (d->K % 128) != 0) return LSTC_E_SHAPE;      // the contraction runs over whole 128-token row blocks of the packs
    P1Params p;
    p.split_stride = splits > 1 ? d->batch_stride_c : 0;
    p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = (float*)d->C;
    p.bias = d->bias; p.res = (const float*)d->residual; p.relu_src = (const float*)d->relu_src;
    p.M = d->M; p.N = d->N; p.ldc = d->ldc; p.ldr = d->ldr; p.ld_relu = d->ld_relu; p.flags = d->flags; p.alpha = d->alpha;
    p.dk = make_drop_key(d->dropout_p, d->dropout_seed);
#ifdef LSTC_TUNING
    p.debug = d->variant >> 4;
#else
    p.debug = 0;
    if (d->variant != 0) return LSTC_E_UNSUPPORTED;
#endif
    p.KBa = (int)p1_kbp(tr ? d->M : d->K);
    p.KBb = (int)p1_kbp(tr ? d->N : d->K);
    p.nsteps = (d->K + 63) / 64;
    p.steps_per_split = (p.nsteps + splits - 1) / splits;
    const int eff_splits = (p.nsteps + p.steps_per_split - 1) / p.steps_per_split;
    const int tilesM = (d->M + 255) / 256;
    p.tilesN = (d->N + 255) / 256;
    constexpr size_t lds = (size_t)2 * P1_BUF * sizeof(bf16_t);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (tr) hipLaunchKernelGGL(gemm_bf16p_kernel<true>, dim3(tilesM * p.tilesN, eff_splits), dim3(NT8), lds, st, p);
    else hipLaunchKernelGGL(gemm_bf16p_kernel<false>, dim3(tilesM * p.tilesN, eff_splits), dim3(NT8), lds, st, p);
    return lstc_launch_status();
}

extern "C" {

int64_t lstc_pack1_bytes(int64_t rows, int64_t K) {
    if (rows <= 0 || K <= 0) return 0;
    return p1_rbp(rows) * p1_kbp(K) * (int64_t)P1_TILE * (int64_t)sizeof(bf16_t) + P1_SLACK;
}

int lstc_pack1(const float* src, int64_t rows, int64_t K, int64_t ld, int32_t k_major, void* dst, void* stream) {
    if (!src || !dst) return LSTC_E_NULL;
    if (rows <= 0 || K <= 0 || ld < (k_major ? rows : K)) return LSTC_E_SHAPE;
    if (!aligned16(dst)) return LSTC_E_ALIGN;
    const int64_t RBp = p1_rbp(rows), KBp = p1_kbp(K);
    if (RBp * KBp > 0x7fffffffLL || rows > 0x7fffffffLL || K > 0x7fffffffLL) return LSTC_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    bf16_t* out = reinterpret_cast<bf16_t*>(dst);
    // every tile of the even-by-even tile grid is written (zeros outside the matrix): the 256-row / 64-k kernel streams whole
    // tile pairs and a garbage k tile would add into real outputs
    if (k_major)
        hipLaunchKernelGGL(pack1_km_kernel, dim3((unsigned)(RBp * KBp)), dim3(256), 0, st, src, (int)rows, (int)K, (long long)ld, out, (int)KBp);
    else
        hipLaunchKernelGGL(pack1_kc_kernel, dim3((unsigned)(RBp * ((KBp + P1_KPB - 1) / P1_KPB))), dim3(256), 0, st, src, (int)rows, (int)K,
                           (long long)ld, out, (int)KBp);
    return lstc_launch_status();
}

}  // extern "C"
