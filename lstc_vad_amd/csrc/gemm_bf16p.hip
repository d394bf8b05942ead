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
#include <cstdlib>
#include <type_traits>
#include "lstc_common.h"

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4v __attribute__((ext_vector_type(4)));

#ifndef P1_NT_S16
#define P1_NT_S16 1                      // NT form on v_mfma_f32_16x16x32_bf16 (0: 32x32x16, A/B builds)
#endif
#ifndef P1_TR_S16
#define P1_TR_S16 0                      // 1: the TR (weight-gradient) form on v_mfma_f32_16x16x32_bf16 too (A/B builds).  Measured, round 6:
                                         // correct (tools/gemm_check check: ALL PASS) but the register allocator no longer fits the loop into
                                         // 256 VGPRs (40 - 58 spills, scratch reloads inside the LOAD segments drain the LDS-DMA queue):
                                         // 2048 x 2048 x 100352 924 TFLOP/s against 1122 on 32x32x16, 6144 x 2048 954 against 1254 - stays off
#endif
#ifndef P1_SPLIT_MAJOR
#define P1_SPLIT_MAJOR 1                 // item order, see set_item (0 / 0: the round-3 order, A/B builds)
#endif
#ifndef P1_GROUP_M
#define P1_GROUP_M 4
#endif
#ifndef P1_RPK_ASM
#define P1_RPK_ASM 1                     // packed residual loads of the pipelined epilogue: 1 = inline asm with hand-counted waits (default);
                                         // 0 = ordinary loads - the compiler then sinks them below the hand-placed waits and spills around
                                         // them (scratch traffic counts in vmcnt too): wrong results with a bias, kept only as a warning
#endif
constexpr int NT8 = 512;                 // 8 waves
constexpr int P1_TILE = 4096;            // bf16 elements of one packed tile (128 rows x 32 k = 8 KB)
constexpr int P1_SLOT = 2048;            // 64 rows x 32 k (4 KB)
constexpr int P1_BUF = 16 * P1_SLOT;     // 64 KB
constexpr int P1_SLACK = 65536;          // bytes after the tiles (TR feature-block over-reads of ragged outputs)

// ---- pack, K-contiguous source [rows, K] (ld): one workgroup per row block and 4 consecutive k tiles.
constexpr int P1_KPB = 4;
__device__ __forceinline__ void pack1_kc_block(const float* __restrict__ x, int rows, int K, long long ld, bf16_t* __restrict__ out, int KBp,
                                               int block) {
    const int kgroups = (KBp + P1_KPB - 1) / P1_KPB;
    const int kb0 = (block % kgroups) * P1_KPB, rb = block / kgroups;
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

__global__ void __launch_bounds__(256) pack1_kc_kernel(const float* __restrict__ x, int rows, int K, long long ld,
                                                       bf16_t* __restrict__ out, int KBp) {
    pack1_kc_block(x, rows, K, ld, out, KBp, (int)blockIdx.x);
}

// ---- pack, k-major source [K, rows] (ld): the packed operand's row index runs along the source's contiguous dimension.
__device__ __forceinline__ void pack1_km_block(const float* __restrict__ x, int rows, int K, long long ld, bf16_t* __restrict__ out, int KBp,
                                               int block) {
    const int kb = block % KBp, rb = block / KBp;
    bf16x8* o = reinterpret_cast<bf16x8*>(out) + (size_t)block * 512;
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
__global__ void __launch_bounds__(256) pack1_km_kernel(const float* __restrict__ x, int rows, int K, long long ld,
                                                       bf16_t* __restrict__ out, int KBp) {
    pack1_km_block(x, rows, K, ld, out, KBp, (int)blockIdx.x);
}

// ---- several operands in ONE launch (lstc_pack1_multi: the weights of a model after an optimizer step - ~35 launches of 10-20 us,
// most of them ramp-up and drain, become one).  The items ride in the kernel arguments; a workgroup finds its item by a wave-uniform
// scan of the block offsets and runs the single-operand code on its local block index: results are those of lstc_pack1, bit for bit.
constexpr int P1_MULTI_MAX = 56;
struct P1Multi {
    const float* src[P1_MULTI_MAX];
    bf16_t* dst[P1_MULTI_MAX];
    long long ld[P1_MULTI_MAX];
    int rows[P1_MULTI_MAX], K[P1_MULTI_MAX], KBp[P1_MULTI_MAX], k_major[P1_MULTI_MAX];
    int first_block[P1_MULTI_MAX + 1];
    int count;
};
__global__ void __launch_bounds__(256) pack1_multi_kernel(const P1Multi b) {
    int t = 0;
    while (t + 1 < b.count && (int)blockIdx.x >= b.first_block[t + 1]) ++t;
    const int block = (int)blockIdx.x - b.first_block[t];
    if (b.k_major[t]) pack1_km_block(b.src[t], b.rows[t], b.K[t], b.ld[t], b.dst[t], b.KBp[t], block);
    else pack1_kc_block(b.src[t], b.rows[t], b.K[t], b.ld[t], b.dst[t], b.KBp[t], block);
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
    int tilesM, tilesN, nsteps, steps_per_split;     // K steps of 64
    int splits, total_items;                 // K splits launched; work items = tiles x splits
    int KBa, KBb;                            // 32-k tiles per 128-row block of the A / B pack (even)
    long long split_stride;                  // elements between the outputs of consecutive K splits (0: atomics into one C)
    int debug;                               // -DLSTC_TUNING builds only: 1 = skip the epilogue (timing ablation), 2 = narrow epilogue
    int vec_epi;                             // 16-B epilogue accesses allowed (N, ld's multiples of 4, pointers 16-B aligned)
    int out_kbp;                             // LSTC_EPI_OUT_PACK: C is an lstc_pack1 buffer of [M, N]; its 32-k tiles per row block (else 0)
    int mask_kbp;                            // LSTC_EPI_RELU_MASK_PACK: relu_src is an lstc_pack1 buffer of [M, N] (else 0)
    int res_kbp;                             // LSTC_EPI_RESIDUAL_PACK: res is an lstc_pack1 buffer of [M, N] (else 0)
};

constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n) only

// PERSISTENT: gridDim.x workgroups (one per CU) walk the work items (output tile x K split) round by round; in a round the
// 32 workgroups of an XCD (blockIdx % 8) take 32 consecutive tiles (N fastest), so their A panels are shared in that XCD's
// L2.  Between two items the LDS-DMA of the NEXT item's first two K steps is issued BEFORE the epilogue of the current one:
// the 822 MB of f32 output per forward GEMM are then in flight while the next tile's first K steps compute.  vmcnt is one
// in-order counter for DMA and stores, so the waits of those first K steps count the epilogue's stores in: the wide epilogue
// issues EXACTLY 32 global_store_dwordx4 per wave through inline asm (full tiles only), which makes "prologue DMA landed"
// = vmcnt(32 + younger DMA) a compile-time constant.  Every other epilogue drains (vmcnt(0)) and starts the next item cold.
// S16 (NT form only): the products run on v_mfma_f32_16x16x32_bf16 instead of 32x32x16 - same FLOP per cycle, same fragment
// bytes, but the chip holds a higher clock on this shape under load (MI355X_MICROARCH.md, DVFS give-back item 7).
// EPK (S16 only): 0 = f32 output / f32 ReLU-mask operand; 1 = the output is written as a packed bf16 operand (LSTC_EPI_OUT_PACK);
// 2 = packed output AND the ReLU mask read from a packed operand (LSTC_EPI_RELU_MASK_PACK); 3 = packed mask, f32 output;
// 4 = packed output AND the residual read from a packed operand (LSTC_EPI_RESIDUAL_PACK: the bf16 activation stream of
// round 5 - the residual sum dropout(f) + x and the input gradient dX + dy never exist in f32).
template <bool TR, bool S16, int EPK = 0>
__global__ void __launch_bounds__(NT8, 2) gemm_bf16p_kernel(const P1Params p) {
    const DropKey dkn = drop_key_now(p.dk);      // graph replays: seed + device offset (lstc_dropout_seed_device)
    // (rounds 2-5 kept the transposed-read form on 32x32x16; round 6: its fragments are read for the 16x16x32 shape too - per 16-lane
    // group one 4-token x 16-feature block per ds_read_b64_tr_b16, group g = the k group 8 g .. 8 g + 7 of the instruction's 32)
    static_assert(EPK == 0 || S16, "packed outputs / masks exist on the pipelined epilogue of the S16 form only");
    constexpr bool OPK = EPK == 1 || EPK == 2 || EPK == 4, MPK = EPK == 2 || EPK == 3, RPK = EPK == 4;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_p1[];
    bf16_t* const smem = smem_p1;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    // work item -> (tile, split); per round the XCD's workgroups own a contiguous run of items
    const int G = gridDim.x, per_xcd = G >> 3;
    const int first = (G & 7) ? (int)blockIdx.x : ((int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3));
    int item = first;
    int mb = 0, nb = 0, kt0 = 0, nkt = 0;
    // item -> (K split, tile).  Consecutive items run side by side on one XCD (32 per round), so they should SHARE operands:
    //   P1_SPLIT_MAJOR: the split index is the slow one (items of one split are consecutive: the tiles of a split share the token
    //     slices of both packs; split-fastest put the four splits of ONE tile side by side, which share nothing - the weight-
    //     gradient form then fetched the X pack once per XCD: 3.6 GB of L2 misses for 0.82 GB of packs);
    //   P1_GROUP_M: inside the tile index, groups of P1_GROUP_M consecutive M panels are walked M fastest, so 32 consecutive tiles
    //     cover P1_GROUP_M panels x 32 / P1_GROUP_M N tiles whatever tilesN is (row-major: 32 / tilesN panels x tilesN tiles -
    //     1.3 x 24 for the fused Q|K|V projection, whose 12-MB weight then streamed through every XCD's L2 each round).
    // Same-box A/B (tools/bf16p_order_ab.sh, round 4): weight gradients 2048 x 2048 x 100352 0.661 -> 0.625 ms, 6144 x 2048 1.789 ->
    // 1.657, 2048 x 4096 1.186 -> 1.113 (FETCH_SIZE x 2 of the first: 3.63 -> 1.22 GB); forward N = 4096 1.517 -> 1.470, N = 6144
    // 2.213 -> 2.123 (FETCH x 2 9.6 -> 4.5 GB); N = 2048 products unchanged (8 N tiles: both orders coincide).
    auto set_item = [&](int w) {
#if P1_SPLIT_MAJOR
        const int ntiles = p.tilesM * p.tilesN;
        const int sp = w / ntiles, tile = w - sp * ntiles;
#else
        const int tile = w / p.splits, sp = w - tile * p.splits;
#endif
#if P1_GROUP_M
        {
            const int per_group = P1_GROUP_M * p.tilesN;
            const int gid = tile / per_group, first_m = gid * P1_GROUP_M;
            const int gsz = min(p.tilesM - first_m, P1_GROUP_M);
            const int loc = tile - gid * per_group;
            mb = first_m + loc % gsz;
            nb = loc / gsz;
        }
#else
        mb = tile / p.tilesN; nb = tile - mb * p.tilesN;
#endif
        kt0 = sp * p.steps_per_split;
        nkt = min(p.nsteps, kt0 + p.steps_per_split) - kt0;
    };

#ifdef LSTC_TUNING
#define P1_ABL_DMA (p.debug & 4)          /* timing ablations (tools/tuning only): no steady-state DMA / no fragment reads */
#define P1_ABL_RD (p.debug & 8)
#define P1_ABL_NOSTORE (p.debug & 16)      /* epilogue arithmetic, no store instruction */
#define P1_ABL_L2STORE (p.debug & 32)      /* every tile stores into rows 0-255 (the output stays in L2) */
#define P1_ABL_HOTDMA (p.debug & 64)       /* every DMA reads K step 0 of tile (0, 0): same instruction stream, always cache hits */
#define P1_ABL_NOXPOSE (p.debug & 128)     /* fast epilogue without the quad transposes / 16-lane exchange (values land in wrong places) */
#define P1_STAMPS (p.debug & 0x2000)       /* round 6: s_memtime stamps per item (tools/gemm_check stamps): where a persistent workgroup's time goes */
#else
#define P1_ABL_DMA 0
#define P1_ABL_RD 0
#define P1_ABL_NOSTORE 0
#define P1_ABL_L2STORE 0
#define P1_ABL_HOTDMA 0
#define P1_ABL_NOXPOSE 0
#define P1_STAMPS 0
#endif
#ifdef LSTC_TUNING
    // stamps live in LDS until the workgroup ends (a global store per stamp would join the hand-counted vmcnt queue): wave 0 writes
    // [item][0..3] = item start / K step 0 landed / K loop done / epilogue done, then copies them to p.relu_src (the tuning harness
    // passes a buffer there and no ReLU-mask flag): [block][0] = {memtime, memrealtime} at start, [block][1] = at end, [block][2 + 4 i + j]
    // (behind the two operand buffers of the dynamic allocation - the LDS-DMA addresses are absolute, a static array would sit at 0)
    unsigned long long* const p1_stamps = reinterpret_cast<unsigned long long*>(smem_p1 + 2 * P1_BUF);
    int p1_item_no = 0;
#define P1_STAMP(j)                                                                                                    \
    do {                                                                                                               \
        if (P1_STAMPS && wave == 0 && p1_item_no < 24) {                                                                \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                \
            if (lane == 0) p1_stamps[4 + 4 * p1_item_no + (j)] = t_;                                                    \
        }                                                                                                              \
    } while (0)
    if (P1_STAMPS && wave == 0) {
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(), r_ = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { p1_stamps[0] = t_; p1_stamps[1] = r_; }
    }
#else
#define P1_STAMP(j) do { } while (0)
#endif
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
        const int KBx = isb ? p.KBb : p.KBa;
        int ob = isb ? nb : mb;
        if (P1_ABL_HOTDMA) { ob = 0; kt = 0; }
        if (TR) return base + ((size_t)(kt >> 1) * KBx + 8 * ob + s) * P1_TILE + (kt & 1) * P1_SLOT;
        return base + ((size_t)(2 * ob + (s >> 2)) * KBx + 2 * kt + (s & 1)) * P1_TILE + ((s >> 1) & 1) * P1_SLOT;
    };
#define P1_DMA_UNIT(u, ktl, buf)                                                                                       \
    do {                                                                                                               \
        if (P1_ABL_DMA) break;                                                                                         \
        const int slot_ = unit_slot(u), pc_ = unit_piece(u);                                                           \
        const bf16_t* g_ = slot_gaddr(slot_, kt0 + (ktl)) + pc_ * 512;                                                 \
        const uint32_t l_ = (uint32_t)(((buf) * P1_BUF + slot_ * P1_SLOT + pc_ * 512) * 2);                            \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" \
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
        if (P1_ABL_RD) return bf16x8{};
        if (TR) return rd_tr(buf + (4 * wr + i) * P1_SLOT, q);
        const bf16_t* s = buf + (2 * (2 * wr + (i >> 1)) + (q >> 1)) * P1_SLOT + (i & 1) * 1024;
        return *reinterpret_cast<const bf16x8*>(s + ((q & 1) ? nt_off1 : nt_off0));
    };
    auto rd_b = [&](const bf16_t* buf, int j, int q) -> bf16x8 {
        if (P1_ABL_RD) return bf16x8{};
        if (TR) return rd_tr(buf + (8 + 2 * wc + j) * P1_SLOT, q);
        const bf16_t* s = buf + (8 + 2 * wc + (q >> 1)) * P1_SLOT + j * 1024;
        return *reinterpret_cast<const bf16x8*>(s + ((q & 1) ? nt_off1 : nt_off0));
    };
    // S16 fragments: lane (l15, c16) reads row l15 of a 16-row tile, 16-B chunk c16 ^ swz of the 32-k slot (16 lanes of a
    // ds_read_b128 group = 16 rows of one chunk column -> 16 distinct 16-B slots, as for the 32-row fragments)
    // ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}
    // (MI355X_MICROARCH.md, LDS): rows 0-3 and 12-15 of k group g travel with rows 4-11 of group g ^ 1.  With the packs' chunk
    // swizzle (row >> 2) & 3, the 16 lanes of every group land on 16 distinct 16-B slots iff the lane groups read the LOGICAL
    // chunks {0, 3, 1, 2} (any k permutation is fine for the product: A and B fragments use the same one); the identity map
    // measured SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE.
    const int l15 = lane & 15, c16 = lane >> 4;
    const int s16_off = l15 * 32 + ((((0x9C >> (2 * c16)) & 3) ^ ((l15 >> 2) & 3)) * 8);      // 0x9C: 2-bit entries 0, 3, 1, 2 for lane groups 0..3
    // S16 + TR: a TR slot holds 64 tokens x 32 features (64-B token rows, chunk XOR (token >> 2) & 3).  The 16x16x32 operand wants, per
    // lane, feature l15 and the tokens 8 c16 .. 8 c16 + 7 of a 32-token half: two ds_read_b64_tr_b16, group c16 reading the 4-token x
    // 16-feature blocks at tokens 8 c16 (+4); lane 4 q + p of a group supplies token row q, features 4 p .. 4 p + 3 of the block.
    // fh = which 16 features of the slot's 32 (it flips bit 1 of the chunk index under the XOR: four per-lane offsets in all)
    const int tq16 = (lane >> 2) & 3, tp16 = lane & 3;
    auto s16tr_at = [&](int fh, int second) -> int {
        const int r = 8 * c16 + tq16 + 4 * second, c = 2 * fh + (tp16 >> 1);
        return r * 32 + ((c ^ ((r >> 2) & 3)) * 8) + (tp16 & 1) * 4;
    };
    const int s16tr_off[2][2] = {{s16tr_at(0, 0), s16tr_at(0, 1)}, {s16tr_at(1, 0), s16tr_at(1, 1)}};
    auto rd_tr16 = [&](const bf16_t* slot, int kk, int fh) -> bf16x8 {           // tokens 32 kk .. +31, features 16 fh .. +15 of the slot
        typedef short short4v __attribute__((ext_vector_type(4)));
        typedef short short8v __attribute__((ext_vector_type(8)));
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(slot + kk * 1024 + s16tr_off[fh][0]));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(slot + kk * 1024 + s16tr_off[fh][1]));
        short8v f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return __builtin_bit_cast(bf16x8, f);
    };
    auto rd_a16 = [&](const bf16_t* buf, int i2, int rt, int kk) -> bf16x8 {     // rows 64 i2 + 16 rt .. +15 of the wave's 128, k tile kk
        if (P1_ABL_RD) return bf16x8{};
        if (TR) return rd_tr16(buf + (4 * wr + 2 * i2 + (rt >> 1)) * P1_SLOT, kk, rt & 1);
        return *reinterpret_cast<const bf16x8*>(buf + (2 * (2 * wr + i2) + kk) * P1_SLOT + rt * 512 + s16_off);
    };
    auto rd_b16 = [&](const bf16_t* buf, int ct4, int kk) -> bf16x8 {            // columns 16 ct4 .. +15 of the wave's 64, k tile kk
        if (P1_ABL_RD) return bf16x8{};
        if (TR) return rd_tr16(buf + (8 + 2 * wc + (ct4 >> 1)) * P1_SLOT, kk, ct4 & 1);
        return *reinterpret_cast<const bf16x8*>(buf + (8 + 2 * wc + kk) * P1_SLOT + ct4 * 512 + s16_off);
    };
    // A: [k16 step][row tile of the current half] (S16: [2 kk + (rt >> 1)][rt & 1]); B: [column half][k16 step] (S16: [jj][2 kk + ct])
    bf16x8 fa[4][2], fb[2][4];
    floatx16 acc[S16 ? 1 : 4][S16 ? 1 : 2];
    floatx4v a4[S16 ? 8 : 1][S16 ? 4 : 1];      // S16: 16x16 accumulators [row tile of the wave's 128][column tile of its 64]

    // K steps 0 and 1 of the current item (mb, nb, kt0, nkt) -> buffers 0 and 1, all four units each: 8 (+8) DMA per wave
    auto issue_head = [&]() {
        P1_DMA_UNIT(0, 0, 0); P1_DMA_UNIT(1, 0, 0); P1_DMA_UNIT(2, 0, 0); P1_DMA_UNIT(3, 0, 0);
        if (nkt > 1) { P1_DMA_UNIT(0, 1, 1); P1_DMA_UNIT(1, 1, 1); P1_DMA_UNIT(2, 1, 1); P1_DMA_UNIT(3, 1, 1); }
    };

#ifndef P1_DMA_IN_COMPUTE
#define P1_DMA_IN_COMPUTE 0
#endif
#ifndef P1_DMA_FIRST
#define P1_DMA_FIRST 0        /* 1: a LOAD segment issues its LDS-DMA unit BEFORE its fragment reads */
#endif
/* MID: statement issued between the two halves of the phase's MFMA burst (P1_DMA_IN_COMPUTE: this phase's LDS-DMA unit, so that
   the LOAD segment holds nothing but the fragment reads) */
#define P1_MMA(ih, jj, rs, ...)                                                                                         \
    do {                                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        if constexpr (S16) {                                                                                           \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                          \
                _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                        \
                    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                    \
                        a4[4 * (ih) + rt][2 * (jj) + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
                            OPK ? fb[rs][2 * kk + ct] : fa[2 * kk + (rt >> 1)][rt & 1],                                 \
                            OPK ? fa[2 * kk + (rt >> 1)][rt & 1] : fb[rs][2 * kk + ct],                                 \
                            (HEAD && kk == 0) ? floatx4v{0.f, 0.f, 0.f, 0.f} : a4[4 * (ih) + rt][2 * (jj) + ct], 0, 0, 0); \
                if (kk == 0) { __builtin_amdgcn_sched_barrier(0); __VA_ARGS__; __builtin_amdgcn_sched_barrier(0); }     \
            }                                                                                                          \
        } else {                                                                                                       \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                             \
                _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                        \
                    acc[S16 ? 0 : 2 * (ih) + ii][S16 ? 0 : jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                 \
                        fa[q][ii], fb[rs][q], (HEAD && q == 0) ? floatx16{} : acc[S16 ? 0 : 2 * (ih) + ii][S16 ? 0 : jj], 0, 0, 0); \
                if (q == 1) { __builtin_amdgcn_sched_barrier(0); __VA_ARGS__; __builtin_amdgcn_sched_barrier(0); }      \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
    } while (0)
/* vmw >= 0: s_waitcnt vmcnt(vmw) before the OPENING barrier.  Waves 4-7 run one barrier behind waves 0-3: the closing barrier of
   waves 0-3 is the opening barrier of waves 4-7, so a wait that must precede a read which waves 0-3 issue right after their
   closing barrier has to sit before the opening one (before the closing one it let waves 0-3 read pieces whose DMA waves 4-7 had
   not waited for yet: a rare mismatch of a weight gradient on a cold first launch). */
#define P1_SYNC_COMPUTE(ih, jj, rs, vmw, ...)                                                                             \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if constexpr ((vmw) >= 0) __builtin_amdgcn_s_waitcnt(vmcnt_imm((vmw) >= 0 ? (vmw) : 0));                       \
        __builtin_amdgcn_s_barrier();                                                                                  \
        __builtin_amdgcn_s_waitcnt(0xC07F);          /* lgkmcnt(0): this phase's fragments are in registers */         \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        P1_MMA(ih, jj, rs, __VA_ARGS__);                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)

    // One K step.  HEAD = K step 0 of an item: K step 1 is already in flight whole (issue_head), and PEND stores of the
    // previous item's epilogue (0 or 32, younger than that DMA) may still be outstanding.
    auto tile = [&](int tl, auto bufc, auto headc, auto pendc) {
        constexpr int CUR = decltype(bufc)::value;
        constexpr bool HEAD = decltype(headc)::value;
        constexpr int PEND = decltype(pendc)::value;
        const bf16_t* cur = smem + CUR * P1_BUF;
        const bool has1 = tl + 1 < nkt, has2 = tl + 2 < nkt;
        // Fragment reads per LOAD segment: 8 (fa rows 0-63) / 4 (fb cols 32-63) / 8 (fa rows 64-127) / 4 (fb cols 0-31 of step
        // t+1, out of the other buffer) - 12 / 4 / 8 / 0 made phase 0's LOAD (x 4 waves on the LDS array, plus latency) longer than
        // the partner's 256-cycle COMPUTE.  The two fb register sets swap roles every K step (the set that held cols 32-63 is
        // free from phase 3 on and receives the next step's cols 0-31): X = set of cols 0-31 this step, Y = the other.
        constexpr int X = CUR, Y = CUR ^ 1;        // an item's K steps alternate CUR = 0, 1, 0, ... from its HEAD step (CUR 0)
        const bf16_t* nxt = smem + (CUR ^ 1) * P1_BUF;
        // ---- phase 0: quadrant (rows first 64, cols first 32)
        if constexpr (HEAD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fb[X][q] = S16 ? rd_b16(cur, q & 1, q >> 1) : rd_b(cur, 0, q);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (P1_DMA_FIRST) { if (!HEAD && has1) P1_DMA_UNIT(2, tl + 1, CUR ^ 1); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa[q][0] = S16 ? rd_a16(cur, 0, 2 * (q & 1), q >> 1) : rd_a(cur, 0, q);
            fa[q][1] = S16 ? rd_a16(cur, 0, 2 * (q & 1) + 1, q >> 1) : rd_a(cur, 1, q);
        }
        if constexpr (P1_DMA_IN_COMPUTE) {
            P1_SYNC_COMPUTE(0, 0, X, -1, if (!HEAD && has1) P1_DMA_UNIT(2, tl + 1, CUR ^ 1));
        } else {
            if constexpr (!P1_DMA_FIRST) { if (!HEAD && has1) P1_DMA_UNIT(2, tl + 1, CUR ^ 1); }
            P1_SYNC_COMPUTE(0, 0, X, -1, (void)0);
        }
        // ---- phase 1: (first 64 rows, second 32 cols)
        if constexpr (P1_DMA_FIRST) { if (!HEAD && has1) P1_DMA_UNIT(3, tl + 1, CUR ^ 1); }
#pragma unroll
        for (int q = 0; q < 4; ++q) fb[Y][q] = S16 ? rd_b16(cur, 2 + (q & 1), q >> 1) : rd_b(cur, 1, q);
        if constexpr (P1_DMA_IN_COMPUTE) {
            P1_SYNC_COMPUTE(0, 1, Y, -1, if (!HEAD && has1) P1_DMA_UNIT(3, tl + 1, CUR ^ 1));
        } else {
            if constexpr (!P1_DMA_FIRST) { if (!HEAD && has1) P1_DMA_UNIT(3, tl + 1, CUR ^ 1); }
            P1_SYNC_COMPUTE(0, 1, Y, -1, (void)0);
        }
        // ---- phase 2: (second 64 rows, second 32 cols).  Before its OPENING barrier: unit U1 of step t+1 (B cols 0-31, issued
        // >= 3 phases ago) has landed - younger operations: U2, U3 of t+1 (4), U0 of t+2 (2), the previous item's stores (HEAD)
        if constexpr (P1_DMA_FIRST) { if (has2) P1_DMA_UNIT(0, tl + 2, CUR); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fa[q][0] = S16 ? rd_a16(cur, 1, 2 * (q & 1), q >> 1) : rd_a(cur, 2, q);
            fa[q][1] = S16 ? rd_a16(cur, 1, 2 * (q & 1) + 1, q >> 1) : rd_a(cur, 3, q);
        }
        if constexpr (P1_DMA_IN_COMPUTE) {      // U0 of step t+2 is issued after the wait: younger ops are U2, U3 of t+1 only
            P1_SYNC_COMPUTE(1, 1, Y, 4 + (HEAD ? PEND : 0), if (has2) P1_DMA_UNIT(0, tl + 2, CUR));
        } else if (has2) {
            if constexpr (!P1_DMA_FIRST) P1_DMA_UNIT(0, tl + 2, CUR);
            P1_SYNC_COMPUTE(1, 1, Y, 6 + (HEAD ? PEND : 0), (void)0);
        } else {
            P1_SYNC_COMPUTE(1, 1, Y, 4 + (HEAD ? PEND : 0), (void)0);
        }
        // ---- phase 3: (second 64 rows, first 32 cols): reads fb cols 0-31 of step t+1 into the set phase 2 just released; K step
        // t+1 must have landed whole before the next step's reads
        if constexpr (P1_DMA_FIRST) { if (has2) P1_DMA_UNIT(1, tl + 2, CUR); }
        if (has1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fb[Y][q] = S16 ? rd_b16(nxt, q & 1, q >> 1) : rd_b(nxt, 0, q);
        }
        if constexpr (P1_DMA_IN_COMPUTE) {      // U1 of step t+2 is issued after the wait: only U0 of t+2 (and the stores) may be open
            if (has2) __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 + (HEAD ? PEND : 0)));
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(HEAD ? PEND : 0));
            P1_SYNC_COMPUTE(1, 0, X, -1, if (has2) P1_DMA_UNIT(1, tl + 2, CUR));
        } else {
            if (has2) {
                if constexpr (!P1_DMA_FIRST) P1_DMA_UNIT(1, tl + 2, CUR);
                __builtin_amdgcn_s_waitcnt(vmcnt_imm(4 + (HEAD ? PEND : 0)));   // everything but U0, U1 of step t+2 (and the stores)
            } else {
                __builtin_amdgcn_s_waitcnt(vmcnt_imm(HEAD ? PEND : 0));
            }
            P1_SYNC_COMPUTE(1, 0, X, -1, (void)0);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    // stores of an item's hand-counted epilogue that may still be outstanding when the next item's K step 0 is waited for: 32 per wave
    // (8-B / 16-B stores of 4 outputs), 16 in the packed-output kernels (16-B stores of 8 outputs, round 6)
    constexpr int PENDN = OPK ? 16 : 32;
    typedef std::integral_constant<int, PENDN> I32;
    typedef std::integral_constant<bool, true> BT;
    typedef std::integral_constant<bool, false> BF;

    const int total = p.total_items;
    if (item >= total) return;
    // (Tried: staggering the workgroups' start per XCD or inside an XCD so that their epilogues do not collide - no gain
    // per XCD, 2-14 % slower inside an XCD: the 64 MB of a round's output drain at the HBM write rate either way.)
    set_item(item);
    issue_head();
    bool pending = false;                 // 32 epilogue stores of the previous item younger than the head DMA
    while (true) {
        P1_STAMP(0);
        // ---- K step 0 landed?  younger ops: K step 1 (8) and the pending stores (32)
        if (nkt > 1) { if (pending) __builtin_amdgcn_s_waitcnt(vmcnt_imm(8 + PENDN)); else __builtin_amdgcn_s_waitcnt(vmcnt_imm(8)); }
        else { if (pending) __builtin_amdgcn_s_waitcnt(vmcnt_imm(PENDN)); else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); }
        __builtin_amdgcn_s_barrier();
        P1_STAMP(1);
        if (wr == 1) __builtin_amdgcn_s_barrier();       // waves 4-7 run one barrier behind: their LOAD beside the partner's COMPUTE
        // (no zeroing pass: in an item's HEAD K step the first MFMA of every accumulator takes the constant 0 as its C operand - round 6,
        // 128 v_mov per wave and item less)
        if (pending) tile(0, I0{}, BT{}, I32{}); else tile(0, I0{}, BT{}, I0{});
        int tl = 1;
        if (tl < nkt) { tile(tl, I1{}, BF{}, I0{}); ++tl; }
        for (; tl + 1 < nkt; tl += 2) {
            tile(tl, I0{}, BF{}, I0{});
            tile(tl + 1, I1{}, BF{}, I0{});
        }
        if (tl < nkt) tile(tl, I0{}, BF{}, I0{});
        if (wr == 0) __builtin_amdgcn_s_barrier();       // balance the stagger: every LDS read of the item is complete
        __builtin_amdgcn_sched_barrier(0);
        P1_STAMP(2);

        // ---- next item: request its first two K steps before this item's epilogue
        const int cmb = mb, cnb = nb, csplit = kt0 / p.steps_per_split;
        const int next = item + G;
        const bool has_next = next < total;
        if (has_next) set_item(next);
        // Hand-counted epilogue (S16, full tile, a next item of >= 2 K steps): every global access of the epilogue - bias,
        // residual / ReLU-mask operand, output - goes through inline asm with counted waits.  Left to the compiler, each
        // `v += residual[...]` gets an s_waitcnt vmcnt(0) (it cannot see the asm stores and the LDS-DMA already in flight), which
        // drains the previous group's store and the next item's head DMA: 32 serialized round trips per tile (measured:
        // 100352 x 2048 x 2048 with a residual 770 TFLOP/s against 1055 without).
        bool fastepi = false;
        floatx4v fbias[2] = {floatx4v{0.f, 0.f, 0.f, 0.f}, floatx4v{0.f, 0.f, 0.f, 0.f}};
        // OPK (packed output, round 6): the products ran with the MFMA operands SWAPPED (B fragment first), so a 16 x 16 accumulator holds,
        // per lane, ONE output row (l15) and four consecutive columns (4 c16 + r) - no quad transposes.  One v_permlane16_swap per
        // register between the two column tiles of a 32-column pack tile leaves every lane with EIGHT consecutive columns = one 16-B
        // chunk of the pack: a wave's 128 x 64 outputs leave as 16 global_store_dwordx4 (was 32 global_store_dwordx2; the epilogue is
        // store-ISSUE bound: tools/r06_bf16p_stamps.sh, 11.4 us per item whatever the flags), operands arrive as 16-B loads of the same
        // shape, ALL issued before the first group is computed (no wait ever covers a store).
        // lane rows c16 = 0 .. 3 of an accumulator end up with the chunks 0, 2, 1, 3 of the 32-column tile:
        const int pk_chunk = (0xD8 >> (2 * c16)) & 3;
        floatx4v pbias[2][2] = {{floatx4v{0.f, 0.f, 0.f, 0.f}, floatx4v{0.f, 0.f, 0.f, 0.f}}, {floatx4v{0.f, 0.f, 0.f, 0.f}, floatx4v{0.f, 0.f, 0.f, 0.f}}};
        if constexpr (OPK) {
            if (p.flags & LSTC_EPI_BIAS) {
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    const float* bp = p.bias + (cnb * 256 + wc * 64 + 32 * cp);                        // wave-uniform
                    const uint32_t bo = (uint32_t)pk_chunk * 32u;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pbias[cp][0]) : "v"(bo), "s"(bp) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(pbias[cp][1]) : "v"(bo), "s"(bp) : "memory");
                }
            }
        }
        if constexpr (S16 && !OPK && !TR) {      // (the TR form writes one tile per workgroup: no pipelined epilogue)
            const int f_ = p.flags;
            fastepi = p.vec_epi && !(p.splits > 1 && p.split_stride == 0) && !(f_ & LSTC_EPI_ACCUM) &&
                      !((f_ & LSTC_EPI_RESIDUAL) && (f_ & LSTC_EPI_RELU_MASK)) && (cmb + 1) * 256 <= p.M && (cnb + 1) * 256 <= p.N;
#ifdef LSTC_TUNING
            if (p.debug & ~(128 | 0x2000)) fastepi = false;
#endif
            if (fastepi && (f_ & LSTC_EPI_BIAS)) {
#pragma unroll
                for (int cp2 = 0; cp2 < 2; ++cp2) {
                    const float* bp = p.bias + (cnb * 256 + wc * 64 + 32 * cp2);                       // wave-uniform
                    const uint32_t bo = ((c16 & 1) * 16 + 4 * (l15 >> 2)) * 4u;          // the lane's columns after the tile exchange
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fbias[cp2]) : "v"(bo), "s"(bp) : "memory");
                }
            }
        }
        if (has_next) issue_head();
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue of (cmb, cnb, csplit) (semantics of gemm_f32.hip)
        bool done = false;
#ifdef LSTC_TUNING
        if (p.debug & 1) {       // timing ablation: keep the accumulators live, store nothing
            float sacc = 0.f;
            if constexpr (S16) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc += (a4[i][j][0] + a4[i][j][1]) + (a4[i][j][2] + a4[i][j][3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc += acc[S16 ? 0 : i][S16 ? 0 : j][r];
            }
            if (sacc == 1.2345e-30f) p.C[0] = sacc;
            done = true; pending = false;
            __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        }
#endif
        const int flags = p.flags;
        const bool atomic = p.splits > 1 && p.split_stride == 0;
        float* const Cz = p.C + (size_t)csplit * p.split_stride;
        const float alpha = p.alpha;
        if constexpr (OPK) {
            if (!done) {
                typedef unsigned uint4v __attribute__((ext_vector_type(4)));
                typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
                const bool f_res = (flags & LSTC_EPI_RESIDUAL) != 0, f_mask = (flags & LSTC_EPI_RELU_MASK) != 0;
                const int hdma = has_next ? (nkt > 1 ? 16 : 8) : 0;                              // LDS-DMA of the next item in flight
                const int nbias = (flags & LSTC_EPI_BIAS) ? 4 : 0;
                // byte offset of the lane's 16-B chunk inside a 16-row x 32-column sub-tile of a pack (64-B rows, chunk XOR (row >> 2) & 3)
                const uint32_t offP = (uint32_t)(l15 * 32 + ((pk_chunk ^ ((l15 >> 2) & 3)) << 3)) * 2u;
                const int urow0 = cmb * 256 + wr * 128, ucol0 = cnb * 256 + wc * 64;
                const uint32_t idx0 = (uint32_t)(urow0 + l15) * (uint32_t)p.N + (uint32_t)(ucol0 + 8 * pk_chunk);
                // f32 operand (residual or ReLU-mask source kept in f32: the f32-activation configuration): row-major, the lane's 8 columns
                const float* aux = f_res ? p.res : f_mask ? p.relu_src : nullptr;
                const int ldx = f_res ? p.ldr : p.ld_relu;
                const uint32_t offX = ((uint32_t)l15 * (uint32_t)ldx + 8u * (uint32_t)pk_chunk) * 4u;
                auto tile_elems = [&](int kbp, int cp, int rt) -> size_t {
                    return ((size_t)(2 * cmb + wr) * kbp + cnb * 8 + wc * 2 + cp) * P1_TILE + rt * 512;
                };
                // one unit u = 8 rt' + ... : (cp = u >> 3, rt = u & 7): the two 16-column accumulators a4[rt][2 cp], a4[rt][2 cp + 1]
                auto unit = [&](int rt, int cp, const uint4v& opk, const floatx4v& o0, const floatx4v& o1) {
                    float e0 = a4[S16 ? rt : 0][S16 ? 2 * cp : 0][0], e1 = a4[S16 ? rt : 0][S16 ? 2 * cp : 0][1];
                    float e2 = a4[S16 ? rt : 0][S16 ? 2 * cp : 0][2], e3 = a4[S16 ? rt : 0][S16 ? 2 * cp : 0][3];
                    float o_0 = a4[S16 ? rt : 0][S16 ? 2 * cp + 1 : 0][0], o_1 = a4[S16 ? rt : 0][S16 ? 2 * cp + 1 : 0][1];
                    float o_2 = a4[S16 ? rt : 0][S16 ? 2 * cp + 1 : 0][2], o_3 = a4[S16 ? rt : 0][S16 ? 2 * cp + 1 : 0][3];
#define P1_SWAP(x, y)                                                                                                    \
                    do {                                                                                                 \
                        const uint2v r_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false); \
                        x = __uint_as_float(r_[0]); y = __uint_as_float(r_[1]);                                           \
                    } while (0)
                    P1_SWAP(e0, o_0); P1_SWAP(e1, o_1); P1_SWAP(e2, o_2); P1_SWAP(e3, o_3);
#undef P1_SWAP
                    float v[8] = {e0, e1, e2, e3, o_0, o_1, o_2, o_3};
                    const float bvv[8] = {pbias[cp][0][0], pbias[cp][0][1], pbias[cp][0][2], pbias[cp][0][3],
                                          pbias[cp][1][0], pbias[cp][1][1], pbias[cp][1][2], pbias[cp][1][3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * alpha + bvv[e];
                    if (flags & LSTC_EPI_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (flags & LSTC_EPI_DROPOUT) {
                        const uint32_t idx = idx0 + (uint32_t)(rt * 16) * (uint32_t)p.N + 32u * (uint32_t)cp;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = drop_keep(idx + e, dkn) ? v[e] * dkn.scale : 0.f;
                    }
                    if constexpr (RPK) {             // bf16 -> f32 is a 16-bit shift: exact
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] += __uint_as_float(opk[e] << 16);
                            v[2 * e + 1] += __uint_as_float(opk[e] & 0xffff0000u);
                        }
                    } else if (f_res) {
                        const float xx[8] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += xx[e];
                    }
                    if constexpr (MPK) {             // bf16 > 0 <=> the element's 16 bits, as the upper half of an int32, are > 0
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] = (int)(opk[e] << 16) > 0 ? v[2 * e] : 0.f;
                            v[2 * e + 1] = (int)(opk[e] & 0xffff0000u) > 0 ? v[2 * e + 1] : 0.f;
                        }
                    } else if (f_mask) {
                        const float xx[8] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = xx[e] > 0.f ? v[e] : 0.f;
                    }
                    uint4v h_;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bf16x2 t2; t2[0] = (bf16_t)v[2 * e]; t2[1] = (bf16_t)v[2 * e + 1];
                        h_[e] = __builtin_bit_cast(unsigned, t2);
                    }
                    bf16_t* ob_ = reinterpret_cast<bf16_t*>(p.C) + tile_elems(p.out_kbp, cp, rt);
                    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(offP), "v"(h_), "s"(ob_) : "memory");
                };
                if constexpr (RPK || MPK) {
                    // 16 packed-operand loads (16 B each), ALL issued first: queue = [bias 4] [head DMA] L0 .. L15 S0 S1 ...; before unit u
                    // the ops younger than L(u) are 15 - u loads and u stores = 15, a constant - and no wait ever covers a store
                    uint4v opk[16];
                    const bf16_t* src = reinterpret_cast<const bf16_t*>(RPK ? p.res : p.relu_src);
                    const int kbp = RPK ? p.res_kbp : p.mask_kbp;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const bf16_t* rb_ = src + tile_elems(kbp, u >> 3, u & 7);
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(opk[u]) : "v"(offP), "s"(rb_) : "memory");
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        __builtin_amdgcn_s_waitcnt(vmcnt_imm(15));
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("" : "+v"(opk[u]));          // nothing computed from the load may move above its wait
                        if (u == 0) { asm volatile("" : "+v"(pbias[0][0]), "+v"(pbias[0][1]), "+v"(pbias[1][0]), "+v"(pbias[1][1])); }
                        unit(u & 7, u >> 3, opk[u], floatx4v{0.f, 0.f, 0.f, 0.f}, floatx4v{0.f, 0.f, 0.f, 0.f});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (aux) {
                    // f32 operand rows: two 16-B loads per unit, one 32-column half (8 units, 16 loads) at a time
                    floatx4v ox[8][2];
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp) {
#pragma unroll
                        for (int rt = 0; rt < 8; ++rt) {
                            const float* ab_ = aux + (size_t)(urow0 + rt * 16) * ldx + (ucol0 + 32 * cp);
                            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ox[rt][0]) : "v"(offX), "s"(ab_) : "memory");
                            asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ox[rt][1]) : "v"(offX), "s"(ab_) : "memory");
                        }
#pragma unroll
                        for (int rt = 0; rt < 8; ++rt) {
                            // younger than this unit's two loads: 2 (7 - rt) loads and rt stores
                            switch (rt) {
                                case 0: __builtin_amdgcn_s_waitcnt(vmcnt_imm(14)); break;
                                case 1: __builtin_amdgcn_s_waitcnt(vmcnt_imm(13)); break;
                                case 2: __builtin_amdgcn_s_waitcnt(vmcnt_imm(12)); break;
                                case 3: __builtin_amdgcn_s_waitcnt(vmcnt_imm(11)); break;
                                case 4: __builtin_amdgcn_s_waitcnt(vmcnt_imm(10)); break;
                                case 5: __builtin_amdgcn_s_waitcnt(vmcnt_imm(9)); break;
                                case 6: __builtin_amdgcn_s_waitcnt(vmcnt_imm(8)); break;
                                default: __builtin_amdgcn_s_waitcnt(vmcnt_imm(7)); break;
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            asm volatile("" : "+v"(ox[rt][0]), "+v"(ox[rt][1]));
                            if (cp == 0 && rt == 0) { asm volatile("" : "+v"(pbias[0][0]), "+v"(pbias[0][1]), "+v"(pbias[1][0]), "+v"(pbias[1][1])); }
                            unit(rt, cp, uint4v{0u, 0u, 0u, 0u}, ox[rt][0], ox[rt][1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
                    if (nbias) {                                  // the bias (issued before the head DMA) has landed; the DMA stays in flight
                        if (hdma == 16) __builtin_amdgcn_s_waitcnt(vmcnt_imm(16));
                        else if (hdma == 8) __builtin_amdgcn_s_waitcnt(vmcnt_imm(8));
                        else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("" : "+v"(pbias[0][0]), "+v"(pbias[0][1]), "+v"(pbias[1][0]), "+v"(pbias[1][1]));
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        unit(u & 7, u >> 3, uint4v{0u, 0u, 0u, 0u}, floatx4v{0.f, 0.f, 0.f, 0.f}, floatx4v{0.f, 0.f, 0.f, 0.f});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                done = true;
                if (has_next) pending = true;
                else { pending = false; __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); }
            }
        }
        if constexpr (S16 && !OPK && !TR) {      // (the TR form writes one tile per workgroup: no pipelined epilogue)
            if (!done && fastepi) {
                // operation order per wave: [bias 2] [head DMA 16] L0 L1 | wait L0 | S0 L2 | wait L1 | S1 L3 | ... (Lb / Sb = the 4
                // operand loads / 4 stores of batch b = column pair b >> 2, row tiles 2 (b & 3), 2 (b & 3) + 1)
                const float* aux = (flags & LSTC_EPI_RESIDUAL) ? p.res : (flags & LSTC_EPI_RELU_MASK) ? p.relu_src : nullptr;
                const int ldx = (flags & LSTC_EPI_RESIDUAL) ? p.ldr : p.ld_relu;
                // Addressing: wave-uniform base in SGPRs (tile, row tile, column pair: scalar arithmetic) + ONE per-lane byte offset
                // shared by all 32 groups (row 4 (c16 & 1) + c4, column 16 (c16 >> 1) + 4 (l15 >> 2) of the group's 8 x 32 block)
                const int c4 = lane & 3;
                // v_permlane16_swap below: lanes 0-15 / 32-47 keep the left tile's rows 0-3 / 8-11, lanes 16-31 / 48-63 receive the right
                // tile's - the two 64-B halves of a 128-B output line then sit in ADJACENT quarter-waves of one store instruction
                // (with the half-wave exchange they were two quarter-waves apart and reached L2 as separate half-line writes:
                // WRITE_SIZE 1.006 GB for an 822-MB output)
                const uint32_t lrow = 8 * (c16 >> 1) + c4, lcol = (c16 & 1) * 16 + 4 * (l15 >> 2);
                const uint32_t offC = (lrow * (uint32_t)p.ldc + lcol) * 4u, offX = (lrow * (uint32_t)ldx + lcol) * 4u;
                const int urow0 = cmb * 256 + wr * 128, ucol0 = cnb * 256 + wc * 64;
                const uint32_t idx0 = (uint32_t)(urow0 + (int)lrow) * (uint32_t)p.N + (uint32_t)(ucol0 + (int)lcol);
                floatx4v ax[2][4];
                // packed [M, N] operand (output / mask): byte offset inside the group's 128-row x 32-k tile, rows +0 / +4
                const uint32_t pch = (lcol & 31) >> 3;
                const uint32_t offP0 = (lrow * 32 + ((pch ^ ((2 * (c16 >> 1)) & 3)) << 3) + (lcol & 7)) * 2u;
                const uint32_t offP1 = ((lrow + 4) * 32 + ((pch ^ ((2 * (c16 >> 1) + 1) & 3)) << 3) + (lcol & 7)) * 2u;
                typedef unsigned uint2v __attribute__((ext_vector_type(2)));
                typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
                uint2v axp[MPK ? 2 : 1][MPK ? 4 : 1];
                uint2v axr[RPK ? 2 : 1][RPK ? 4 : 1];      // RPK: 4 bf16 of the packed residual per group, hand-counted like ax
                const int hdma = has_next ? (nkt > 1 ? 16 : 8) : 0;                              // LDS-DMA of the next item in flight
#define P1_FE_LOAD(b, set)                                                                                              \
                do {                                                                                                    \
                    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                   \
                        if constexpr (MPK) {                                                                            \
                            const bf16_t* mb_ = reinterpret_cast<const bf16_t*>(p.relu_src) +                             \
                                ((size_t)(2 * cmb + wr) * p.mask_kbp + cnb * 8 + wc * 2 + ((b) >> 2)) * P1_TILE + (2 * ((b) & 3) + (g_ >> 1)) * 512; \
                            /* an ordinary load: the compiler waits for it itself (its count ignores the asm stores, so the wait  \
                               also covers the previous batch's stores - the price of never having these registers in flight     \
                               behind its back; the f32 operands below stay hand-counted) */                                      \
                            axp[MPK ? set : 0][MPK ? g_ : 0] = *reinterpret_cast<const uint2v*>(reinterpret_cast<const char*>(mb_) + ((g_ & 1) ? offP1 : offP0)); \
                        } else if constexpr (RPK) {   /* the tile address is wave-uniform (SGPRs), the lane offset one of two VGPRs */ \
                            const bf16_t* rb_ = reinterpret_cast<const bf16_t*>(p.res) +                                  \
                                ((size_t)(2 * cmb + wr) * p.res_kbp + cnb * 8 + wc * 2 + ((b) >> 2)) * P1_TILE + (2 * ((b) & 3) + (g_ >> 1)) * 512; \
                            if constexpr (P1_RPK_ASM) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(axr[RPK ? set : 0][RPK ? g_ : 0]) : "v"((g_ & 1) ? offP1 : offP0), "s"(rb_) : "memory"); \
                            else axr[RPK ? set : 0][RPK ? g_ : 0] = *reinterpret_cast<const uint2v*>(reinterpret_cast<const char*>(rb_) + ((g_ & 1) ? offP1 : offP0)); \
                        } else {                                                                                        \
                            const float* ab_ = aux + (size_t)(urow0 + (2 * ((b) & 3) + (g_ >> 1)) * 16 + 4 * (g_ & 1)) * ldx + (ucol0 + 32 * ((b) >> 2)); \
                            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ax[set][g_]) : "v"(offX), "s"(ab_) : "memory"); \
                        }                                                                                               \
                    }                                                                                                   \
                } while (0)
                auto xpose2 = [&](float& v0, float& v1, float& v2, float& v3) {
                    const bool b1 = (c4 & 2) != 0, b0 = (c4 & 1) != 0;
                    int s0 = __float_as_int(b1 ? v0 : v2), s1 = __float_as_int(b1 ? v1 : v3);
                    float r0 = __int_as_float(__builtin_amdgcn_mov_dpp(s0, 0x4E, 0xF, 0xF, true));
                    float r1 = __int_as_float(__builtin_amdgcn_mov_dpp(s1, 0x4E, 0xF, 0xF, true));
                    if (b1) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
                    s0 = __float_as_int(b0 ? v0 : v1); s1 = __float_as_int(b0 ? v2 : v3);
                    r0 = __int_as_float(__builtin_amdgcn_mov_dpp(s0, 0xB1, 0xF, 0xF, true));
                    r1 = __int_as_float(__builtin_amdgcn_mov_dpp(s1, 0xB1, 0xF, 0xF, true));
                    if (b0) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                };
#define P1_FE_GROUP(v0, v1, v2, v3, rt, hf, cp, bv, av, mv, rv)                                                              \
                do {                                                                                                    \
                    float4 v = make_float4((v0) * alpha + (bv)[0], (v1) * alpha + (bv)[1], (v2) * alpha + (bv)[2], (v3) * alpha + (bv)[3]); \
                    if (flags & LSTC_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); } \
                    if (flags & LSTC_EPI_DROPOUT) {                                                                      \
                        const uint32_t idx = idx0 + (uint32_t)((rt) * 16 + 4 * (hf)) * (uint32_t)p.N + 32u * (cp);        \
                        v.x = drop_keep(idx, dkn) ? v.x * dkn.scale : 0.f;                                             \
                        v.y = drop_keep(idx + 1, dkn) ? v.y * dkn.scale : 0.f;                                         \
                        v.z = drop_keep(idx + 2, dkn) ? v.z * dkn.scale : 0.f;                                         \
                        v.w = drop_keep(idx + 3, dkn) ? v.w * dkn.scale : 0.f;                                         \
                    }                                                                                                   \
                    if (RPK || (flags & LSTC_EPI_RESIDUAL)) {                                                            \
                        if constexpr (RPK) {   /* bf16 -> f32 is a 16-bit shift: exact */                               \
                            v.x += __uint_as_float((rv)[0] << 16); v.y += __uint_as_float((rv)[0] & 0xffff0000u);        \
                            v.z += __uint_as_float((rv)[1] << 16); v.w += __uint_as_float((rv)[1] & 0xffff0000u);        \
                        } else { v.x += (av)[0]; v.y += (av)[1]; v.z += (av)[2]; v.w += (av)[3]; }                       \
                    }                                                                                                   \
                    if (flags & LSTC_EPI_RELU_MASK) {                                                                    \
                        if constexpr (MPK) {   /* bf16 > 0 <=> the element's 16 bits, as the upper half of an int32, are > 0 */ \
                            v.x = (int)((mv)[0] << 16) > 0 ? v.x : 0.f; v.y = (int)((mv)[0] & 0xffff0000u) > 0 ? v.y : 0.f; \
                            v.z = (int)((mv)[1] << 16) > 0 ? v.z : 0.f; v.w = (int)((mv)[1] & 0xffff0000u) > 0 ? v.w : 0.f; \
                        } else {                                                                                        \
                            v.x = (av)[0] > 0.f ? v.x : 0.f; v.y = (av)[1] > 0.f ? v.y : 0.f;                             \
                            v.z = (av)[2] > 0.f ? v.z : 0.f; v.w = (av)[3] > 0.f ? v.w : 0.f;                             \
                        }                                                                                               \
                    }                                                                                                   \
                    if constexpr (OPK) {   /* the output IS the next product's packed operand: 4 bf16 = 8 B per group */   \
                        bf16x4 h_; h_[0] = (bf16_t)v.x; h_[1] = (bf16_t)v.y; h_[2] = (bf16_t)v.z; h_[3] = (bf16_t)v.w;    \
                        bf16_t* ob_ = reinterpret_cast<bf16_t*>(p.C) +                                                    \
                            ((size_t)(2 * cmb + wr) * p.out_kbp + cnb * 8 + wc * 2 + (cp)) * P1_TILE + (rt) * 512;         \
                        asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" :: "v"((hf) ? offP1 : offP0), "v"(h_), "s"(ob_) : "memory"); \
                    } else {                                                                                            \
                        float* cb_ = Cz + (size_t)(urow0 + (rt) * 16 + 4 * (hf)) * p.ldc + (ucol0 + 32 * (cp));          \
                        const floatx4v sv_ = {v.x, v.y, v.z, v.w};                                                       \
                        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(offC), "v"(sv_), "s"(cb_) : "memory"); \
                    }                                                                                                   \
                } while (0)
                // RPK: the launcher admits the variant only with LSTC_EPI_RESIDUAL set - no runtime branch around the loads (with one
                // the register allocator spilled the first batch's in-flight destinations: tools/isa_guard.py)
                if (RPK || aux) { P1_FE_LOAD(0, 0); }
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    if (RPK || aux) {
                        if (b < 7) { P1_FE_LOAD(b + 1, (b + 1) & 1); }
                        if (b == 0 || b == 7) __builtin_amdgcn_s_waitcnt(vmcnt_imm(4)); else __builtin_amdgcn_s_waitcnt(vmcnt_imm(8));
                    } else if (b == 0 && (flags & LSTC_EPI_BIAS)) {                               // the head DMA stays in flight
                        if (hdma == 16) __builtin_amdgcn_s_waitcnt(vmcnt_imm(16));
                        else if (hdma == 8) __builtin_amdgcn_s_waitcnt(vmcnt_imm(8));
                        else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // the asm loads' outputs count as "ready" for the compiler from the load on: pass them through an empty asm
                    // AFTER the wait, so that nothing computed from them (a mask compare does not depend on the accumulators) can
                    // be moved ahead of it
                    if (b == 0) { asm volatile("" : "+v"(fbias[0]), "+v"(fbias[1])); }
#pragma unroll
                    for (int g_ = 0; g_ < 4; ++g_) {
                        if constexpr (RPK && P1_RPK_ASM) asm volatile("" : "+v"(axr[RPK ? (b & 1) : 0][RPK ? g_ : 0]));
                        else if constexpr (!MPK) asm volatile("" : "+v"(ax[b & 1][g_]));
                    }
                    const int cp2 = b >> 2;
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const int rt = 2 * (b & 3) + r2;
                        float x0 = a4[S16 ? rt : 0][S16 ? 2 * cp2 : 0][0], x1 = a4[S16 ? rt : 0][S16 ? 2 * cp2 : 0][1];
                        float x2 = a4[S16 ? rt : 0][S16 ? 2 * cp2 : 0][2], x3 = a4[S16 ? rt : 0][S16 ? 2 * cp2 : 0][3];
                        float y0 = a4[S16 ? rt : 0][S16 ? 2 * cp2 + 1 : 0][0], y1 = a4[S16 ? rt : 0][S16 ? 2 * cp2 + 1 : 0][1];
                        float y2 = a4[S16 ? rt : 0][S16 ? 2 * cp2 + 1 : 0][2], y3 = a4[S16 ? rt : 0][S16 ? 2 * cp2 + 1 : 0][3];
                        if (!P1_ABL_NOXPOSE) {
                        xpose2(x0, x1, x2, x3);
                        xpose2(y0, y1, y2, y3);
                        }
                        typedef unsigned uint2v __attribute__((ext_vector_type(2)));
#define P1_SWAP(x, y)                                                                                                    \
                        do {                                                                                             \
                            const uint2v r_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false); \
                            x = __uint_as_float(r_[0]); y = __uint_as_float(r_[1]);                                       \
                        } while (0)
                        if (!P1_ABL_NOXPOSE) { P1_SWAP(x0, y0); P1_SWAP(x1, y1); P1_SWAP(x2, y2); P1_SWAP(x3, y3); }
#undef P1_SWAP
                        P1_FE_GROUP(x0, x1, x2, x3, rt, 0, cp2, fbias[cp2], ax[b & 1][2 * r2], axp[MPK ? (b & 1) : 0][MPK ? 2 * r2 : 0],
                                    axr[RPK ? (b & 1) : 0][RPK ? 2 * r2 : 0]);
                        P1_FE_GROUP(y0, y1, y2, y3, rt, 1, cp2, fbias[cp2], ax[b & 1][2 * r2 + 1], axp[MPK ? (b & 1) : 0][MPK ? 2 * r2 + 1 : 0],
                                    axr[RPK ? (b & 1) : 0][RPK ? 2 * r2 + 1 : 0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef P1_FE_LOAD
#undef P1_FE_GROUP
                done = true;
                if (has_next) pending = true;
                else { pending = false; __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); }
            }
        }
        if constexpr (EPK != 0) done = true;      // the launcher admits packed outputs / masks only where every tile takes the path above
        if (!done && p.vec_epi && !atomic) {
            // wide epilogue: a 32x32 accumulator holds, per lane, ONE column and 16 rows (4 consecutive rows per register
            // group); a 4x4 transpose inside each quad of lanes (DPP quad_perm, no LDS) turns a register group into 4
            // consecutive columns of one row, so every lane moves 16 B and a wave-instruction covers 8 rows x 128 B full lines
            // instead of 2 rows x 128 B with four times the instructions (cdna_hip_programming.md T21).
            const bool full = (cmb + 1) * 256 <= p.M && (cnb + 1) * 256 <= p.N;       // wave-uniform
            const int c4 = lane & 3, q8 = l31 >> 2;
            auto xpose = [&](float& v0, float& v1, float& v2, float& v3) {
                const bool b1 = (c4 & 2) != 0, b0 = (c4 & 1) != 0;
                int s0 = __float_as_int(b1 ? v0 : v2), s1 = __float_as_int(b1 ? v1 : v3);
                float r0 = __int_as_float(__builtin_amdgcn_mov_dpp(s0, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
                float r1 = __int_as_float(__builtin_amdgcn_mov_dpp(s1, 0x4E, 0xF, 0xF, true));
                if (b1) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
                s0 = __float_as_int(b0 ? v0 : v1); s1 = __float_as_int(b0 ? v2 : v3);
                r0 = __int_as_float(__builtin_amdgcn_mov_dpp(s0, 0xB1, 0xF, 0xF, true));            // quad_perm [1,0,3,2]
                r1 = __int_as_float(__builtin_amdgcn_mov_dpp(s1, 0xB1, 0xF, 0xF, true));
                if (b0) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
            };
            // one output group: 4 consecutive columns of one row held by this lane -> epilogue math -> ONE 16-B store
#define P1_EPI_STORE(v0, v1, v2, v3, row, col, cok, bv)                                                                  \
            do {                                                                                                         \
                if (!full && (!(cok) || (row) >= p.M)) break;                                                              \
                float4 v = make_float4((v0) * alpha + (bv).x, (v1) * alpha + (bv).y, (v2) * alpha + (bv).z, (v3) * alpha + (bv).w); \
                if (flags & LSTC_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); } \
                if (flags & LSTC_EPI_DROPOUT) {                                                                            \
                    const uint32_t idx = (uint32_t)(row) * (uint32_t)p.N + (uint32_t)(col);                                \
                    v.x = drop_keep(idx, dkn) ? v.x * dkn.scale : 0.f;                                                   \
                    v.y = drop_keep(idx + 1, dkn) ? v.y * dkn.scale : 0.f;                                               \
                    v.z = drop_keep(idx + 2, dkn) ? v.z * dkn.scale : 0.f;                                               \
                    v.w = drop_keep(idx + 3, dkn) ? v.w * dkn.scale : 0.f;                                               \
                }                                                                                                        \
                if (flags & LSTC_EPI_RESIDUAL) {                                                                           \
                    const float4 r = *reinterpret_cast<const float4*>(p.res + (size_t)(row) * p.ldr + (col));              \
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;                                                        \
                }                                                                                                        \
                if (flags & LSTC_EPI_RELU_MASK) {                                                                          \
                    const float4 m = *reinterpret_cast<const float4*>(p.relu_src + (size_t)(row) * p.ld_relu + (col));     \
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f; \
                }                                                                                                        \
                float* cp = Cz + (size_t)(row) * p.ldc + (col);                                                            \
                if (flags & LSTC_EPI_ACCUM) { const float4 o = *reinterpret_cast<const float4*>(cp); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; } \
                if (P1_ABL_NOSTORE) { const floatx4v sv = {v.x, v.y, v.z, v.w}; asm volatile("" :: "v"(sv), "v"(cp)); break; }       \
                if (P1_ABL_L2STORE) cp = p.C + ((size_t)(row & 255) * p.ldc + (col));                                       \
                if (P1_ABL_NOSTORE) { const floatx4v sv = {v.x, v.y, v.z, v.w}; asm volatile("" :: "v"(sv), "v"(cp)); break; }       \
                if (P1_ABL_L2STORE) cp = p.C + ((size_t)((row) & 255) * p.ldc + (col));                                     \
                if (full) {      /* exactly one store instruction per group: 32 per wave, counted by the next item's waits */ \
                    const floatx4v sv = {v.x, v.y, v.z, v.w};                                                              \
                    /* s_nop 1: the store reads its 16 B of data registers after issue (cdna_hip_programming.md 5.7 item 1) */ \
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(cp), "v"(sv) : "memory");           \
                } else {                                                                                                 \
                    *reinterpret_cast<float4*>(cp) = v;                                                                    \
                }                                                                                                        \
            } while (0)
            if constexpr (S16) {
                // a 16x16 accumulator gives, after the quad transpose, 16 rows x 64 B per wave: half a cache line per row.  Two
                // neighbouring column tiles are therefore exchanged across the wave halves (v_permlane32_swap: lanes 32-63 of the
                // left tile's registers <-> lanes 0-31 of the right tile's), after which lanes 0-31 hold rows 0-7 of BOTH tiles'
                // left... of the LEFT tile and lanes 32-63 rows 0-7 of the RIGHT tile in the first register set (rows 8-15 in the
                // second): every store instruction again writes 8 rows x 128 B full lines.
#pragma unroll
                for (int cp2 = 0; cp2 < 2; ++cp2) {
                    const int col = cnb * 256 + wc * 64 + (2 * cp2 + (c16 >> 1)) * 16 + 4 * (l15 >> 2);
                    const bool cok = col < p.N;
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if ((flags & LSTC_EPI_BIAS) && cok) bv = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
                    for (int rt = 0; rt < 8; ++rt) {
                        float x0 = a4[S16 ? rt : 0][S16 ? 2 * cp2 : 0][0], x1 = a4[S16 ? rt : 0][S16 ? 2 * cp2 : 0][1];
                        float x2 = a4[S16 ? rt : 0][S16 ? 2 * cp2 : 0][2], x3 = a4[S16 ? rt : 0][S16 ? 2 * cp2 : 0][3];
                        float y0 = a4[S16 ? rt : 0][S16 ? 2 * cp2 + 1 : 0][0], y1 = a4[S16 ? rt : 0][S16 ? 2 * cp2 + 1 : 0][1];
                        float y2 = a4[S16 ? rt : 0][S16 ? 2 * cp2 + 1 : 0][2], y3 = a4[S16 ? rt : 0][S16 ? 2 * cp2 + 1 : 0][3];
                        xpose(x0, x1, x2, x3);
                        xpose(y0, y1, y2, y3);
                        typedef unsigned uint2v __attribute__((ext_vector_type(2)));
#define P1_SWAP(x, y)                                                                                                    \
                        do {                                                                                             \
                            const uint2v r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false); \
                            x = __uint_as_float(r_[0]); y = __uint_as_float(r_[1]);                                       \
                        } while (0)
                        P1_SWAP(x0, y0); P1_SWAP(x1, y1); P1_SWAP(x2, y2); P1_SWAP(x3, y3);
#undef P1_SWAP
                        const int row = cmb * 256 + wr * 128 + rt * 16 + 4 * (c16 & 1) + c4;
                        P1_EPI_STORE(x0, x1, x2, x3, row, col, cok, bv);
                        P1_EPI_STORE(y0, y1, y2, y3, row + 8, col, cok, bv);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = cnb * 256 + wc * 64 + j * 32 + 4 * q8;
                    const bool cok = col < p.N;                                   // N % 4 == 0: all four columns or none
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if ((flags & LSTC_EPI_BIAS) && cok) bv = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v0 = acc[S16 ? 0 : i][S16 ? 0 : j][4 * g], v1 = acc[S16 ? 0 : i][S16 ? 0 : j][4 * g + 1];
                            float v2 = acc[S16 ? 0 : i][S16 ? 0 : j][4 * g + 2], v3 = acc[S16 ? 0 : i][S16 ? 0 : j][4 * g + 3];
                            xpose(v0, v1, v2, v3);
                            const int row = cmb * 256 + wr * 128 + i * 32 + 4 * h + 8 * g + c4;
                            P1_EPI_STORE(v0, v1, v2, v3, row, col, cok, bv);
                        }
                    }
                }
            }
#undef P1_EPI_STORE
            done = true;
            if (full && has_next) pending = true;
            else { pending = false; __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); }
        }
        if (!done) {
            constexpr int NCG = S16 ? 4 : 2, NRT = S16 ? 8 : 4, NR = S16 ? 4 : 16;
#pragma unroll
            for (int j = 0; j < NCG; ++j) {
                const int col = cnb * 256 + wc * 64 + (S16 ? j * 16 + l15 : j * 32 + l31);
                if (col >= p.N) continue;
                const float bv = (flags & LSTC_EPI_BIAS) ? p.bias[col] : 0.f;
#pragma unroll
                for (int i = 0; i < NRT; ++i) {
                    const int rbase = cmb * 256 + wr * 128 + (S16 ? i * 16 + 4 * c16 : i * 32 + 4 * h);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row >= p.M) continue;
                        float v;
                        if constexpr (S16) v = a4[i][j][r] * alpha; else v = acc[S16 ? 0 : i][S16 ? 0 : j][r] * alpha;
                        float* cp = Cz + (size_t)row * p.ldc + col;
                        if (atomic) {
                            atomicAdd(cp, v);
                            continue;
                        }
                        v += bv;
                        if (flags & LSTC_EPI_RELU) v = fmaxf(v, 0.f);
                        if (flags & LSTC_EPI_DROPOUT) {
                            const uint32_t idx = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;
                            v = drop_keep(idx, dkn) ? v * dkn.scale : 0.f;
                        }
                        if (flags & LSTC_EPI_RESIDUAL) v += p.res[(size_t)row * p.ldr + col];
                        if (flags & LSTC_EPI_RELU_MASK) v = p.relu_src[(size_t)row * p.ld_relu + col] > 0.f ? v : 0.f;
                        if (flags & LSTC_EPI_ACCUM) v += *cp;
                        *cp = v;
                    }
                }
            }
            pending = false;
            __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        }
        __builtin_amdgcn_sched_barrier(0);
        P1_STAMP(3);
#ifdef LSTC_TUNING
        ++p1_item_no;
#endif
        if (!has_next) break;
        item = next;
    }
#ifdef LSTC_TUNING
    if (P1_STAMPS && wave == 0) {
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(), r_ = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { p1_stamps[2] = t_; p1_stamps[3] = r_; }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.relu_src)) + (size_t)blockIdx.x * 100;
        for (int i = lane; i < 100; i += 64) dst[i] = (i < 4 + 4 * p1_item_no) ? p1_stamps[i] : 0ull;
    }
#endif
#undef P1_DMA_UNIT
#undef P1_MMA
#undef P1_SYNC_COMPUTE
}

// ---- QUARTER tiles for the tail round (round 6, second session).  The persistent kernel above pays a launch's last, partly filled
// round of 256 workgroups in full: 3136 tiles of the N = 2048 products are 12.25 rounds paid as 13, the 64 tiles of a 2048 x 2048
// product keep 64 of 256 CUs busy.  The launcher therefore hands the tiles behind the last WHOLE round (tile index >= tile0, same
// tile order as set_item above) to this kernel as four 128 x 128 quarter items each, one workgroup per item: 64 tail tiles = 256
// workgroups of a quarter of the work.
//   * Every output element sees the same K order and the same MFMA chain as in the 256 x 256 tile (one v_mfma_f32_16x16x32_bf16 per
//     16 x 16 block and 32-k tile, kk = 0 before kk = 1, operand order as in the main kernel): results are bit-identical to the
//     one-kernel product (tests/test_act16_gpu.py::test_quarter_tail_*, against d->variant = LSTC_VARIANT_NO_QTAIL).
//   * Producer / consumer waves: a 128 x 128 item has MFMA work for four waves of 64 x 64 (16 ds_read_b128 per 32 MFMAs and K step;
//     eight waves of 64 x 32 would read 12 per 16).  Waves 0-3 (one per SIMD) compute; waves 4-7 own the LDS-DMA
//     stream and its vmcnt - so the consumers' epilogue can use ordinary loads and stores, nothing hand-counted.
//   * A K step of 64 is FOUR whole pack tiles (A and B, k tiles 2 s and 2 s + 1 of the item's 128-row blocks): 8 KB contiguous
//     each, copied verbatim - one producer wave per tile, 8 x 1 KB.  Q_NS stages of 32 KB; the DMA of step s + Q_NS - 1 is issued when
//     step s - 1's stage falls free, fragment registers are double-buffered (reads of step s + 1 beside the MFMAs of step s),
//     ONE s_barrier per K step.
#ifndef P1_QTAIL_ROUNDS
#define P1_QTAIL_ROUNDS 2                 // rounds of quarter items a launch's tail may take (LSTC_P1_QTAIL overrides at run time; 0 = off)
#endif
#ifndef P1_Q_NS
#define P1_Q_NS 4                         // 5 (all 160 KB of the CU, one more K step to land) measured: no change - 2048 x 2048 x 2048 0.024 ms, K = 4096
#endif                                    // 0.046 either way.  The item is not latency-bound: a 16-tile product (64 workgroups, 192 CUs idle) takes the
                                          // same 0.66 us per K step as a 64-tile one - 32 KB per step and CU = 48 GB/s per CU, the rate the 256 x 256
                                          // loop's 64 KB per 1.33 us comes to as well: the CU's LDS-DMA intake (profiles/r06_qtail5_ab.txt)
constexpr int Q_NS = P1_Q_NS;             // stages (32 KB each); the DMA of a K step has Q_NS - 2 steps to land
static_assert(Q_NS == 4 || Q_NS == 5, "quarter items: four or five stages");
constexpr int Q_STAGE = 4 * P1_TILE;      // A k0 | A k1 | B k0 | B k1
template <int EPK>
__global__ void __launch_bounds__(NT8, 2) gemm_bf16p_q_kernel(const P1Params p, const int tile0) {
    const DropKey dkn = drop_key_now(p.dk);
    constexpr bool OPK = EPK == 1 || EPK == 2 || EPK == 4, MPK = EPK == 2, RPK = EPK == 4;
    static_assert(EPK == 0 || EPK == 1 || EPK == 2 || EPK == 4, "quarter items: f32 output, packed output, + packed mask, + packed residual");
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_p1[];
    bf16_t* const smem = smem_p1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int pw = wave & 3;
    // items sharing a tile (and tiles sharing a panel) sit on one XCD: blockIdx % 8 -> a contiguous run of items
    const int nblk = gridDim.x;
    const int item = (nblk & 7) ? (int)blockIdx.x : ((int)(blockIdx.x & 7) * (nblk >> 3) + (int)(blockIdx.x >> 3));
    const int tile = tile0 + (item >> 2), qr = (item >> 1) & 1, qc = item & 1;
    int mb, nb;
#if P1_GROUP_M
    {
        const int per_group = P1_GROUP_M * p.tilesN;
        const int gid = tile / per_group, first_m = gid * P1_GROUP_M;
        const int gsz = min(p.tilesM - first_m, P1_GROUP_M);
        const int loc = tile - gid * per_group;
        mb = first_m + loc % gsz;
        nb = loc / gsz;
    }
#else
    mb = tile / p.tilesN; nb = tile - mb * p.tilesN;
#endif
    const int nkt = p.nsteps;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    // producer wave pw copies pack tile pw of a stage: 0 / 1 = A (row block 2 mb + qr), k tiles 2 s / 2 s + 1; 2 / 3 = B (2 nb + qc)
    const bf16_t* const src0 = (pw >= 2 ? p.B + (size_t)(2 * nb + qc) * p.KBb * P1_TILE : p.A + (size_t)(2 * mb + qr) * p.KBa * P1_TILE) +
                               (size_t)(pw & 1) * P1_TILE;
    auto q_issue = [&](int s) {
        const bf16_t* g_ = src0 + (size_t)(2 * s) * P1_TILE;
        const uint32_t l_ = (uint32_t)(((s % Q_NS) * Q_STAGE + pw * P1_TILE) * 2);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                     :: "v"(lane_off), "s"(g_), "s"(l_) : "memory");
        const bf16_t* g2_ = g_ + 2048;
        const uint32_t l2_ = l_ + 4096u;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                     :: "v"(lane_off), "s"(g2_), "s"(l2_) : "memory");
    };
    // consumer wave pw = (w2r, w2c): rows 64 w2r .. +63, columns 64 w2c .. +63 of the item; fragments as in the main kernel (s16_off)
    const int w2r = pw >> 1, w2c = pw & 1;
    const int l15 = lane & 15, c16 = lane >> 4;
    const int s16_off = l15 * 32 + ((((0x9C >> (2 * c16)) & 3) ^ ((l15 >> 2) & 3)) * 8);
    const int a_off = (64 * w2r) * 32 + s16_off, b_off = 2 * P1_TILE + (64 * w2c) * 32 + s16_off;
    bf16x8 fa0[2][4], fb0[2][4], fa1[2][4], fb1[2][4];
    floatx4v qa[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) qa[i][j] = floatx4v{0.f, 0.f, 0.f, 0.f};
    auto q_read = [&](int s, bf16x8 (&fa_)[2][4], bf16x8 (&fb_)[2][4]) {
        const bf16_t* stg = smem + (s % Q_NS) * Q_STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) fa_[kk][rt] = *reinterpret_cast<const bf16x8*>(stg + kk * P1_TILE + a_off + rt * 512);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fb_[kk][ct] = *reinterpret_cast<const bf16x8*>(stg + kk * P1_TILE + b_off + ct * 512);
        }
    };
    auto q_mma = [&](const bf16x8 (&fa_)[2][4], const bf16x8 (&fb_)[2][4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    qa[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(OPK ? fb_[kk][ct] : fa_[kk][rt], OPK ? fa_[kk][rt] : fb_[kk][ct],
                                                                         qa[rt][ct], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // K step s, as the two kinds of wave see it: the producers' DMA of step s + 1 has landed (younger: step s + 2), everyone meets (ONE
    // s_barrier per step, the same count on both sides), then the producers refill the stage step s - 1 left (every consumer's reads
    // of it completed before its MFMAs of step s - 1) and the consumers read step s + 1 beside the MFMAs of step s.  Two separate
    // loops: with one loop and a branch inside, every fragment register is a phi of "unchanged" and "new" at the join and the
    // allocator spills around the copies (481 spilled registers).
#define Q_BARRIER()                                                                                                     \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
    if (producer) {
        // Q_NS - 1 steps in flight: the DMA of step s + Q_NS - 1 is issued when step s - 1's stage falls free and has Q_NS - 2 steps to land
#pragma unroll
        for (int s0 = 0; s0 < Q_NS - 1; ++s0)
            if (s0 < nkt) q_issue(s0);
        {   // step 0 landed: younger = min(Q_NS - 2, nkt - 1) steps of 8 pieces
            const int y = min(Q_NS - 2, nkt - 1);
            if (y >= 3) __builtin_amdgcn_s_waitcnt(vmcnt_imm(24));
            else if (y == 2) __builtin_amdgcn_s_waitcnt(vmcnt_imm(16));
            else if (y == 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(8));
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        }
        Q_BARRIER();
        for (int s = 0; s < nkt; ++s) {
            // step s + 1 landed: younger = the steps s + 2 .. s + Q_NS - 2 that exist
            const int y = min(Q_NS - 3, nkt - 2 - s);
            if (y >= 2) __builtin_amdgcn_s_waitcnt(vmcnt_imm(16));
            else if (y == 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(8));
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
            Q_BARRIER();
            if (s + Q_NS - 1 < nkt) q_issue(s + Q_NS - 1);
        }
        return;       // (the item is the workgroup's only one: no barrier follows)
    }
    Q_BARRIER();
    q_read(0, fa0, fb0);
    {
        // (reads behind the last step are unconditional: a stage nobody writes, values nobody uses)
        int s = 0;
        for (; s + 1 < nkt; s += 2) {
            Q_BARRIER();
            q_read(s + 1, fa1, fb1);
            q_mma(fa0, fb0);
            Q_BARRIER();
            q_read(s + 2, fa0, fb0);
            q_mma(fa1, fb1);
        }
        if (s < nkt) {
            Q_BARRIER();
            q_mma(fa0, fb0);
        }
    }
#undef Q_BARRIER

    // ---- epilogue of the item: rows urow0 .. +63, columns ucol0 .. +63 per consumer wave (semantics and operation order of the main kernel)
    const int flags = p.flags;
    const float alpha = p.alpha;
    const int urow0 = mb * 256 + qr * 128 + w2r * 64, ucol0 = nb * 256 + qc * 128 + w2c * 64;
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    if constexpr (OPK) {
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        const bool f_res = (flags & LSTC_EPI_RESIDUAL) != 0, f_mask = (flags & LSTC_EPI_RELU_MASK) != 0;
        // swapped operands: a lane holds ONE row (l15) and, after the 16-lane exchange, eight consecutive columns = chunk pk_chunk
        const int pk_chunk = (0xD8 >> (2 * c16)) & 3;
        const uint32_t offP = (uint32_t)(l15 * 32 + ((pk_chunk ^ ((l15 >> 2) & 3)) << 3)) * 2u;
        const uint32_t idx0 = (uint32_t)(urow0 + l15) * (uint32_t)p.N + (uint32_t)(ucol0 + 8 * pk_chunk);
        const float* aux = f_res ? p.res : f_mask ? p.relu_src : nullptr;
        const int ldx = f_res ? p.ldr : p.ld_relu;
        auto tile_elems = [&](int kbp, int cp, int rt) -> size_t {
            return ((size_t)(2 * mb + qr) * kbp + nb * 8 + qc * 4 + w2c * 2 + cp) * P1_TILE + (w2r * 4 + rt) * 512;
        };
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            floatx4v b0 = floatx4v{0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (flags & LSTC_EPI_BIAS) {
                const float* bp = p.bias + (ucol0 + 32 * cp + 8 * pk_chunk);
                b0 = *reinterpret_cast<const floatx4v*>(bp);
                b1 = *reinterpret_cast<const floatx4v*>(bp + 4);
            }
            const float bvv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                float e0 = qa[rt][2 * cp][0], e1 = qa[rt][2 * cp][1], e2 = qa[rt][2 * cp][2], e3 = qa[rt][2 * cp][3];
                float o_0 = qa[rt][2 * cp + 1][0], o_1 = qa[rt][2 * cp + 1][1], o_2 = qa[rt][2 * cp + 1][2], o_3 = qa[rt][2 * cp + 1][3];
#define P1_SWAP(x, y)                                                                                                    \
                do {                                                                                                     \
                    const uint2v r_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false); \
                    x = __uint_as_float(r_[0]); y = __uint_as_float(r_[1]);                                               \
                } while (0)
                P1_SWAP(e0, o_0); P1_SWAP(e1, o_1); P1_SWAP(e2, o_2); P1_SWAP(e3, o_3);
#undef P1_SWAP
                float v[8] = {e0, e1, e2, e3, o_0, o_1, o_2, o_3};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * alpha + bvv[e];
                if (flags & LSTC_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (flags & LSTC_EPI_DROPOUT) {
                    const uint32_t idx = idx0 + (uint32_t)(rt * 16) * (uint32_t)p.N + 32u * (uint32_t)cp;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = drop_keep(idx + e, dkn) ? v[e] * dkn.scale : 0.f;
                }
                uint4v opk = uint4v{0u, 0u, 0u, 0u};
                if constexpr (RPK || MPK) {
                    const bf16_t* src = reinterpret_cast<const bf16_t*>(RPK ? p.res : p.relu_src) + tile_elems(RPK ? p.res_kbp : p.mask_kbp, cp, rt);
                    opk = *reinterpret_cast<const uint4v*>(reinterpret_cast<const char*>(src) + offP);
                }
                float xx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if ((!RPK && f_res) || (!MPK && !f_res && f_mask)) {      // the f32 per-element operand (never the packed one's pointer)
                    const float* ab_ = aux + (size_t)(urow0 + rt * 16 + l15) * ldx + (ucol0 + 32 * cp + 8 * pk_chunk);
                    const floatx4v o0 = *reinterpret_cast<const floatx4v*>(ab_), o1 = *reinterpret_cast<const floatx4v*>(ab_ + 4);
                    xx[0] = o0[0]; xx[1] = o0[1]; xx[2] = o0[2]; xx[3] = o0[3]; xx[4] = o1[0]; xx[5] = o1[1]; xx[6] = o1[2]; xx[7] = o1[3];
                }
                if constexpr (RPK) {             // bf16 -> f32 is a 16-bit shift: exact
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(opk[e] << 16);
                        v[2 * e + 1] += __uint_as_float(opk[e] & 0xffff0000u);
                    }
                } else if (f_res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += xx[e];
                }
                if constexpr (MPK) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = (int)(opk[e] << 16) > 0 ? v[2 * e] : 0.f;
                        v[2 * e + 1] = (int)(opk[e] & 0xffff0000u) > 0 ? v[2 * e + 1] : 0.f;
                    }
                } else if (f_mask) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = xx[e] > 0.f ? v[e] : 0.f;
                }
                uint4v h_;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf16x2 t2; t2[0] = (bf16_t)v[2 * e]; t2[1] = (bf16_t)v[2 * e + 1];
                    h_[e] = __builtin_bit_cast(unsigned, t2);
                }
                bf16_t* ob_ = reinterpret_cast<bf16_t*>(p.C) + tile_elems(p.out_kbp, cp, rt);
                *reinterpret_cast<uint4v*>(reinterpret_cast<char*>(ob_) + offP) = h_;
            }
        }
    } else {
        // f32 output: quad transposes + the half-wave exchange of the main kernel's wide epilogue - a lane stores 4 consecutive columns
        const int c4 = lane & 3;
        auto xpose = [&](float& v0, float& v1, float& v2, float& v3) {
            const bool b1 = (c4 & 2) != 0, b0 = (c4 & 1) != 0;
            int s0 = __float_as_int(b1 ? v0 : v2), s1 = __float_as_int(b1 ? v1 : v3);
            float r0 = __int_as_float(__builtin_amdgcn_mov_dpp(s0, 0x4E, 0xF, 0xF, true));
            float r1 = __int_as_float(__builtin_amdgcn_mov_dpp(s1, 0x4E, 0xF, 0xF, true));
            if (b1) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
            s0 = __float_as_int(b0 ? v0 : v1); s1 = __float_as_int(b0 ? v2 : v3);
            r0 = __int_as_float(__builtin_amdgcn_mov_dpp(s0, 0xB1, 0xF, 0xF, true));
            r1 = __int_as_float(__builtin_amdgcn_mov_dpp(s1, 0xB1, 0xF, 0xF, true));
            if (b0) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
        };
        auto store4 = [&](float v0, float v1, float v2, float v3, int row, int col, const float4& bv) {
            if (col >= p.N || row >= p.M) return;
            float4 v = make_float4(v0 * alpha + bv.x, v1 * alpha + bv.y, v2 * alpha + bv.z, v3 * alpha + bv.w);
            if (flags & LSTC_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (flags & LSTC_EPI_DROPOUT) {
                const uint32_t idx = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;
                v.x = drop_keep(idx, dkn) ? v.x * dkn.scale : 0.f;
                v.y = drop_keep(idx + 1, dkn) ? v.y * dkn.scale : 0.f;
                v.z = drop_keep(idx + 2, dkn) ? v.z * dkn.scale : 0.f;
                v.w = drop_keep(idx + 3, dkn) ? v.w * dkn.scale : 0.f;
            }
            if (flags & LSTC_EPI_RESIDUAL) {
                const float4 r = *reinterpret_cast<const float4*>(p.res + (size_t)row * p.ldr + col);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (flags & LSTC_EPI_RELU_MASK) {
                const float4 m = *reinterpret_cast<const float4*>(p.relu_src + (size_t)row * p.ld_relu + col);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            float* cp = p.C + (size_t)row * p.ldc + col;
            if (flags & LSTC_EPI_ACCUM) { const float4 o = *reinterpret_cast<const float4*>(cp); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(cp) = v;
        };
#pragma unroll
        for (int cp2 = 0; cp2 < 2; ++cp2) {
            const int col = ucol0 + (2 * cp2 + (c16 >> 1)) * 16 + 4 * (l15 >> 2);
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((flags & LSTC_EPI_BIAS) && col < p.N) bv = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                float x0 = qa[rt][2 * cp2][0], x1 = qa[rt][2 * cp2][1], x2 = qa[rt][2 * cp2][2], x3 = qa[rt][2 * cp2][3];
                float y0 = qa[rt][2 * cp2 + 1][0], y1 = qa[rt][2 * cp2 + 1][1], y2 = qa[rt][2 * cp2 + 1][2], y3 = qa[rt][2 * cp2 + 1][3];
                xpose(x0, x1, x2, x3);
                xpose(y0, y1, y2, y3);
#define P1_SWAP(x, y)                                                                                                    \
                do {                                                                                                     \
                    const uint2v r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false); \
                    x = __uint_as_float(r_[0]); y = __uint_as_float(r_[1]);                                               \
                } while (0)
                P1_SWAP(x0, y0); P1_SWAP(x1, y1); P1_SWAP(x2, y2); P1_SWAP(x3, y3);
#undef P1_SWAP
                const int row = urow0 + rt * 16 + 4 * (c16 & 1) + c4;
                store4(x0, x1, x2, x3, row, col, bv);
                store4(y0, y1, y2, y3, row + 8, col, bv);
            }
        }
    }
}

inline int64_t p1_rbp(int64_t rows) { const int64_t rb = (rows + 127) / 128; return rb + (rb & 1); }
inline int64_t p1_kbp(int64_t K) { const int64_t kb = (K + 31) / 32; return kb + (kb & 1); }

}  // namespace

// Packed-operand bf16 GEMM behind lstc_gemm (dtype LSTC_BF16P): d->A / d->B point to lstc_pack1 outputs.
__attribute__((visibility("hidden"))) int lstc_gemm_bf16p_impl(const LstcGemmDesc* d, hipStream_t st) {
    if (!d->A || !d->B || !d->C) return LSTC_E_NULL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (!(d->flags & LSTC_EPI_OUT_PACK) && d->ldc < d->N)) return LSTC_E_SHAPE;
    if (d->batch > 1) return LSTC_E_UNSUPPORTED;
    if ((d->flags & LSTC_EPI_BIAS) && !d->bias) return LSTC_E_NULL;
    if ((d->flags & LSTC_EPI_RESIDUAL) && (!d->residual || (!(d->flags & LSTC_EPI_RESIDUAL_PACK) && d->ldr < d->N))) return LSTC_E_NULL;
    if ((d->flags & LSTC_EPI_RELU_MASK) && (!d->relu_src || (!(d->flags & LSTC_EPI_RELU_MASK_PACK) && d->ld_relu < d->N))) return LSTC_E_NULL;
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
    p.M = d->M; p.N = d->N; p.ldc = d->ldc; p.ldr = d->ldr; p.ld_relu = d->ld_relu; p.alpha = d->alpha;
    p.flags = d->flags & ~(LSTC_EPI_OUT_PACK | LSTC_EPI_RELU_MASK_PACK | LSTC_EPI_RESIDUAL_PACK);
    p.out_kbp = (d->flags & LSTC_EPI_OUT_PACK) ? (int)p1_kbp(d->N) : 0;
    p.mask_kbp = (d->flags & LSTC_EPI_RELU_MASK_PACK) ? (int)p1_kbp(d->N) : 0;
    p.res_kbp = (d->flags & LSTC_EPI_RESIDUAL_PACK) ? (int)p1_kbp(d->N) : 0;
    // a packed residual comes with a packed output only (the bf16 activation stream), never with a ReLU mask
    if (p.res_kbp && (!(d->flags & LSTC_EPI_RESIDUAL) || !p.out_kbp || (d->flags & LSTC_EPI_RELU_MASK))) return LSTC_E_UNSUPPORTED;
    if (p.out_kbp || p.mask_kbp) {
        // packed outputs / mask operands exist on the pipelined epilogue only: NT form on whole 256 x 256 tiles, no K split
#ifdef LSTC_TUNING
        const int tile_variant = d->variant & 15;          // the bits above select timing ablations / stamps
#else
        const int tile_variant = d->variant & ~LSTC_VARIANT_NO_QTAIL;
#endif
        if (tr || splits > 1 || !P1_NT_S16 || d->M % 256 || d->N % 256 || (d->flags & LSTC_EPI_ACCUM) || tile_variant != 0 ||
            (p.mask_kbp && (!(d->flags & LSTC_EPI_RELU_MASK) || (d->flags & LSTC_EPI_RESIDUAL)))) return LSTC_E_UNSUPPORTED;
        if (p.out_kbp && !aligned16(d->C)) return LSTC_E_ALIGN;
        // one per-element operand stream per epilogue (the hand-counted loads): residual OR ReLU mask, never both
        if (p.out_kbp && (d->flags & LSTC_EPI_RESIDUAL) && (d->flags & LSTC_EPI_RELU_MASK)) return LSTC_E_UNSUPPORTED;
    }
    p.dk = make_drop_key(d->dropout_p, d->dropout_seed);
    // d->variant: 0 = default; LSTC_VARIANT_NO_QTAIL = the same product without the quarter-tile tail kernel (one persistent launch: the
    // reference of the bitwise tests).  Tuning builds: bits 0-3 tile variant, bits 4.. timing ablations / stamps
    const bool no_qtail = (d->variant & LSTC_VARIANT_NO_QTAIL) != 0;
    const int variant_ = d->variant & ~LSTC_VARIANT_NO_QTAIL;
#ifdef LSTC_TUNING
    p.debug = variant_ >> 4;
#else
    p.debug = 0;
    if (variant_ != 0) return LSTC_E_UNSUPPORTED;
#endif
    p.vec_epi = (d->N % 4 == 0) && (d->ldc % 4 == 0) && aligned16(d->C) && (p.split_stride % 4 == 0) &&
                (!(d->flags & LSTC_EPI_BIAS) || aligned16(d->bias)) &&
                (!(d->flags & LSTC_EPI_RESIDUAL) || (aligned16(d->residual) && (p.res_kbp || d->ldr % 4 == 0))) &&
                (!(d->flags & LSTC_EPI_RELU_MASK) || (aligned16(d->relu_src) && (p.mask_kbp || d->ld_relu % 4 == 0)));
    if ((p.out_kbp || p.mask_kbp) && !p.vec_epi) return LSTC_E_ALIGN;
#ifdef LSTC_TUNING
    if (p.debug & 2) p.vec_epi = 0;
#endif
    p.KBa = (int)p1_kbp(tr ? d->M : d->K);
    p.KBb = (int)p1_kbp(tr ? d->N : d->K);
    p.nsteps = (d->K + 63) / 64;
    p.steps_per_split = (p.nsteps + splits - 1) / splits;
    const int eff_splits = (p.nsteps + p.steps_per_split - 1) / p.steps_per_split;
    const int tilesM = (d->M + 255) / 256;
    p.tilesM = tilesM;
    p.tilesN = (d->N + 255) / 256;
    p.splits = eff_splits;
    p.total_items = tilesM * p.tilesN * eff_splits;
#ifdef LSTC_TUNING
    constexpr size_t lds = (size_t)2 * P1_BUF * sizeof(bf16_t) + 1024;       // + the stamp array of the tuning build
#else
    constexpr size_t lds = (size_t)2 * P1_BUF * sizeof(bf16_t);
#endif
    constexpr size_t q_lds = (size_t)Q_NS * Q_STAGE * sizeof(bf16_t);
    // per device: the CU count sizes the persistent grid and the 128-KB dynamic-LDS opt-in is a per-device kernel attribute
    static std::atomic<int> n_cu_dev[64];
    static LstcDevOnce setup;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    const int dev_ = setup.begin();
    if (dev_ >= 0) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev_) == hipSuccess) n = prop.multiProcessorCount;
        n_cu_dev[dev_ & 63].store(n > 0 ? n : 256, std::memory_order_relaxed);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<true, P1_TR_S16 != 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<false, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<false, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel<false, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_q_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)q_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_q_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)q_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_q_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)q_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_q_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)q_lds);
        setup.end(dev_);
    }
    const int n_cu = n_cu_dev[cur & 63].load(std::memory_order_relaxed);
    // persistent: one workgroup per CU (128 KB of LDS each).  (Tried, round 5: whole rounds on fewer workgroups - 3136 items of the
    // N = 2048 products are 12.25 rounds of 256 but exactly 14 of 224.  A round takes the same 55 - 56 us either way, so 14 of them
    // lose: 0.732 -> 0.774 ms at K = 2048, 1.303 -> 1.350 at K = 4096, step 38.3 -> 39.2 ms.  The chip is not power-limited at this
    // granularity: idle CUs buy the busy ones nothing.)
    // Quarter-tile tail (gemm_bf16p_q_kernel): the tiles behind the last whole round of n_cu workgroups run as four 128 x 128 items
    // each when that fills the chip better - at most P1_QTAIL_ROUNDS rounds of quarter items (a quarter item takes ~0.35 of a tile's
    // time: two rounds of them still beat one round of tiles, three do not).  NT form on 16x16x32, no K split, 16-B epilogue accesses.
    const int epk_ = p.res_kbp ? 4 : p.out_kbp ? (p.mask_kbp ? 2 : 1) : (p.mask_kbp ? 3 : 0);
    int q_tiles = 0;
    if (!tr && P1_NT_S16 && eff_splits == 1 && p.vec_epi && epk_ != 3 && !no_qtail && variant_ == 0 && n_cu > 0) {
        static const int q_rounds = [] {
            const char* e = getenv("LSTC_P1_QTAIL");            // 0 = off (A/B runs), 1 .. 3 = rounds of quarter items allowed
            const int v = e ? atoi(e) : P1_QTAIL_ROUNDS;
            return v < 0 ? 0 : v > 3 ? 3 : v;
        }();
        const int tail = p.total_items % n_cu;
        if (tail > 0 && 4 * tail <= q_rounds * n_cu) { q_tiles = tail; p.total_items -= tail; }
    }
    const int grid = p.total_items < n_cu ? p.total_items : n_cu;
    if (q_tiles) {
        const int tile0 = p.total_items;
        if (grid > 0) {
            const int epk = epk_;
            if (epk == 0) hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 0>), dim3(grid), dim3(NT8), lds, st, p);
            else if (epk == 1) hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 1>), dim3(grid), dim3(NT8), lds, st, p);
            else if (epk == 2) hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 2>), dim3(grid), dim3(NT8), lds, st, p);
            else hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 4>), dim3(grid), dim3(NT8), lds, st, p);
            const int rc = lstc_launch_status();
            if (rc) return rc;
        }
        if (epk_ == 0) hipLaunchKernelGGL((gemm_bf16p_q_kernel<0>), dim3(4 * q_tiles), dim3(NT8), q_lds, st, p, tile0);
        else if (epk_ == 1) hipLaunchKernelGGL((gemm_bf16p_q_kernel<1>), dim3(4 * q_tiles), dim3(NT8), q_lds, st, p, tile0);
        else if (epk_ == 2) hipLaunchKernelGGL((gemm_bf16p_q_kernel<2>), dim3(4 * q_tiles), dim3(NT8), q_lds, st, p, tile0);
        else hipLaunchKernelGGL((gemm_bf16p_q_kernel<4>), dim3(4 * q_tiles), dim3(NT8), q_lds, st, p, tile0);
        return lstc_launch_status();
    }
    if (tr) hipLaunchKernelGGL((gemm_bf16p_kernel<true, P1_TR_S16 != 0>), dim3(grid), dim3(NT8), lds, st, p);
    else if (P1_NT_S16) {
        const int epk = p.res_kbp ? 4 : p.out_kbp ? (p.mask_kbp ? 2 : 1) : (p.mask_kbp ? 3 : 0);
        if (epk == 0) hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 0>), dim3(grid), dim3(NT8), lds, st, p);
        else if (epk == 1) hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 1>), dim3(grid), dim3(NT8), lds, st, p);
        else if (epk == 2) hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 2>), dim3(grid), dim3(NT8), lds, st, p);
        else if (epk == 4) hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 4>), dim3(grid), dim3(NT8), lds, st, p);
        else hipLaunchKernelGGL((gemm_bf16p_kernel<false, true, 3>), dim3(grid), dim3(NT8), lds, st, p);
    }
    else hipLaunchKernelGGL((gemm_bf16p_kernel<false, false>), dim3(grid), dim3(NT8), lds, st, p);
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

int lstc_pack1_multi(const LstcPackItem* items, int32_t count, void* stream) {
    if (!items) return LSTC_E_NULL;
    if (count <= 0) return LSTC_E_SHAPE;
    for (int32_t i = 0; i < count; ++i) {
        const LstcPackItem& it = items[i];
        if (!it.src || !it.dst) return LSTC_E_NULL;
        if (it.rows <= 0 || it.K <= 0 || it.ld < (it.k_major ? it.rows : it.K)) return LSTC_E_SHAPE;
        if (!aligned16(it.dst)) return LSTC_E_ALIGN;
        if (p1_rbp(it.rows) * p1_kbp(it.K) > 0x7fffffffLL || it.rows > 0x7fffffffLL || it.K > 0x7fffffffLL) return LSTC_E_RANGE;
    }
    for (int32_t base = 0; base < count; base += P1_MULTI_MAX) {
        P1Multi b;
        b.count = count - base < P1_MULTI_MAX ? count - base : P1_MULTI_MAX;
        int64_t blocks = 0;
        for (int i = 0; i < b.count; ++i) {
            const LstcPackItem& it = items[base + i];
            const int64_t RBp = p1_rbp(it.rows), KBp = p1_kbp(it.K);
            b.src[i] = it.src; b.dst[i] = reinterpret_cast<bf16_t*>(it.dst); b.ld[i] = it.ld;
            b.rows[i] = (int)it.rows; b.K[i] = (int)it.K; b.KBp[i] = (int)KBp; b.k_major[i] = it.k_major ? 1 : 0;
            b.first_block[i] = (int)blocks;
            blocks += it.k_major ? RBp * KBp : RBp * ((KBp + P1_KPB - 1) / P1_KPB);
            if (blocks > 0x7fffffffLL) return LSTC_E_RANGE;
        }
        for (int i = b.count; i <= P1_MULTI_MAX; ++i) b.first_block[i] = (int)blocks;
        hipLaunchKernelGGL(pack1_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
        const int rc = lstc_launch_status();
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
