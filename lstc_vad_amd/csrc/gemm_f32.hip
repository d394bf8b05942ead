// Exact-f32 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
// C[M,N] = epi(alpha * op(A) * op(B)), all three operand layouts the training step needs:
//   NT  X * W^T      forward of every nn.Linear             (A K-contig, B K-contig)
//   NN  dY * W       input gradient                         (A K-contig, B N-contig)
//   TN  dY^T * X     weight gradient, K = tokens (~1e5)     (A M-contig, B N-contig)
//
// Design (MI355X_MICROARCH / cdna_hip_programming 3, 5):
//  * v_mfma_f32_32x32x2_f32: the only exact-f32 matrix instruction; 64 cycles per issue per SIMD,
//    so one wave per SIMD with >=4 independent accumulators already saturates the pipe and every
//    other instruction (LDS reads, global prefetch, epilogue of the co-resident workgroup) hides
//    behind it.  Each wave owns a (BM/WGM)x(BN/WGN) tile = TMxTN accumulators of 32x32.
//  * K order is free inside a dot product, so lane half h (lane>>5) takes k = 16h..16h+15 of each
//    32-deep K tile instead of the interleaved {h, h+2, ...}: a K-contiguous operand row is then
//    read from LDS with 4 ds_read_b128 per 32x32x32 block instead of 16 ds_read_b32.
//  * LDS images: K-contiguous operands as [rows][36] floats (144-B rows: 16-B aligned for
//    ds_write_b128 / ds_read_b128, 36 = 4*9 with 9 odd -> every 16-lane b128 group hits 16 distinct
//    16-B slots: conflict-free); M/N-contiguous operands as [32][rows] read with ds_read_b32
//    (32 consecutive floats per half-wave: conflict-free).
//  * Register-staged double buffering: global loads of K tile t+1 are issued before the MFMAs of
//    tile t and written to the other LDS buffer after them; one barrier per K tile.
//  * XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs, so ids are remapped
//    (bijectively) to give each XCD a contiguous run of tiles with the N index fastest: an XCD's L2
//    then holds the A panel its tiles share.
//  * Epilogue in the accumulator layout (col = lane&31, row = (r&3)+8(r>>2)+4(lane>>5)): every
//    wave-level load/store is two 128-B row segments.
#include "lstc_common.h"
#include <type_traits>

#ifndef LSTC_F32_GROUP_M
#define LSTC_F32_GROUP_M 8          /* grouped tile order inside an XCD's run, see the tile map (0: row-major, the order of rounds 1-3) */
#endif
namespace {

constexpr int BK = 32;
constexpr int LDK = 36;   // padded K stride of K-contiguous LDS images
#ifndef LSTC_KCT
#define LSTC_KCT 0
#endif
// LSTC_KCT = 1: K-contiguous operands are TRANSPOSED while they are written to LDS ([32][rows+1] image, 4 ds_write_b32 per
// staged float4, conflict-free because rows+1 = 1 mod 32) and read back like k-major operands with ds_read_b32.
constexpr bool KCT = LSTC_KCT != 0;

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* res;
    const float* relu_src;
    int M, N, K, lda, ldb, ldc, ldr, ld_relu;
    int flags;
    float alpha;
    DropKey dk;
    int tilesM, tilesN, batch;
    int ktiles, ktiles_per_split;
    long long batch_stride_a, batch_stride_b, batch_stride_c;   // elements between consecutive problems of a batch (grid.z)
    unsigned int a_bytes, b_bytes;   // operand extents for the buffer descriptors of PIPE 5 (operands < 4 GiB)
    int debug;     // timing-only ablations (tools/gemm_check): 1 = no global loads in the loop, 2 = no LDS writes, 4 = no barrier
    int row_off;   // first row of this launch inside the caller's matrix (dropout counter of a row-split product)
    int epi_f4;    // 0: scalar epilogue; 1 / 2: float4 epilogue allowed, without / with ONE per-element operand (epilogue_f4)
};

// Stages one operand tile (R rows/cols x 32 k) global -> registers -> LDS.
//  KC = true : operand stored [R_total][K] (K contiguous); LDS image [R][LDK].
//  KC = false: operand stored [K][R_total] (R contiguous); LDS image [32][R].
template <int R, int NT, bool KC, bool VEC>
struct Stager {
    static constexpr int NV = R * 8 / NT;
    static_assert(NV >= 1 && (R * 8) % NT == 0, "tile/threads mismatch");
    float4 v[NV];

    // Rows/cols outside the matrix are CLAMPED to the last valid one instead of predicated: the accumulators they
    // feed belong to output rows/cols that the epilogue never stores, and the main loop stays branch-free.
    // Only the K tail needs zero fill (CHECK_K = true, used for the last K tile when K % 32 != 0).
    template <bool CHECK_K>
    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int r0, int r_total, int k0, int K) {
        const int t = threadIdx.x;
        if (KC) {
            const int c = (t & 7) * 4;
            const int k = k0 + c;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = min(r0 + (t >> 3) + i * (NT / 8), r_total - 1);
                const float* p = base + (size_t)r * ld + k;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (VEC) {
                    if (!CHECK_K || k < K) x = *reinterpret_cast<const float4*>(p);
                } else {
                    if (!CHECK_K || k + 0 < K) x.x = p[0];
                    if (!CHECK_K || k + 1 < K) x.y = p[1];
                    if (!CHECK_K || k + 2 < K) x.z = p[2];
                    if (!CHECK_K || k + 3 < K) x.w = p[3];
                }
                v[i] = x;
            }
        } else {
            constexpr int CPR = R / 4;          // float4 columns per k row
            const int c = r0 + (t % CPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = k0 + t / CPR + i * (NT / CPR);
                const float* p = base + (size_t)(CHECK_K ? min(k, K - 1) : k) * ld;
                float4 x;
                if (VEC) {
                    x = *reinterpret_cast<const float4*>(p + min(c, r_total - 4));
                } else {
                    x.x = p[min(c + 0, r_total - 1)];
                    x.y = p[min(c + 1, r_total - 1)];
                    x.z = p[min(c + 2, r_total - 1)];
                    x.w = p[min(c + 3, r_total - 1)];
                }
                if (CHECK_K && k >= K) x = make_float4(0.f, 0.f, 0.f, 0.f);
                v[i] = x;
            }
        }
    }

    // Single-element forms for the hand-interleaved pipeline (PIPE 3): one global load / one LDS write per call.
    __device__ __forceinline__ void load_one(int i, const float* __restrict__ base, int ld, int r0, int r_total, int k0) {
        const int t = threadIdx.x;
        if (KC) {
            const int r = min(r0 + (t >> 3) + i * (NT / 8), r_total - 1);
            const float* p = base + (size_t)r * ld + k0 + (t & 7) * 4;
            if (VEC) v[i] = *reinterpret_cast<const float4*>(p);
            else v[i] = make_float4(p[0], p[1], p[2], p[3]);
        } else {
            constexpr int CPR = R / 4;
            const int c = r0 + (t % CPR) * 4;
            const float* p = base + (size_t)(k0 + t / CPR + i * (NT / CPR)) * ld;
            if (VEC) v[i] = *reinterpret_cast<const float4*>(p + min(c, r_total - 4));
            else v[i] = make_float4(p[min(c, r_total - 1)], p[min(c + 1, r_total - 1)], p[min(c + 2, r_total - 1)],
                                    p[min(c + 3, r_total - 1)]);
        }
    }
    // Per-lane byte offset of staged element i at K tile 0 (rows clamped); constant over the K loop, so the loop's loads
    // need no vector address arithmetic: buffer_load voffset = this, soffset = K-tile byte offset (scalar).
    __device__ __forceinline__ uint32_t voffset(int i, int ld, int r0, int r_total) const {
        const int t = threadIdx.x;
        if (KC) {
            const int r = min(r0 + (t >> 3) + i * (NT / 8), r_total - 1);
            return (uint32_t)(((size_t)r * ld + (t & 7) * 4) * sizeof(float));
        }
        constexpr int CPR = R / 4;
        const int c = min(r0 + (t % CPR) * 4, r_total - 4);
        return (uint32_t)(((size_t)(t / CPR + i * (NT / CPR)) * ld + c) * sizeof(float));
    }
    __device__ __forceinline__ void load_buf(int i, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
        static_assert(VEC, "buffer path is for 16-B aligned operands");
        typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        v[i] = __builtin_bit_cast(float4, x);
    }
    __device__ __forceinline__ void store_one(int i, float* __restrict__ lds) const {
        const int t = threadIdx.x;
        if (KC) {
            if (KCT) {
                const int r = (t >> 3) + i * (NT / 8), c = (t & 7) * 4;
                lds[(c + 0) * (R + 1) + r] = v[i].x; lds[(c + 1) * (R + 1) + r] = v[i].y;
                lds[(c + 2) * (R + 1) + r] = v[i].z; lds[(c + 3) * (R + 1) + r] = v[i].w;
            } else {
                *reinterpret_cast<float4*>(lds + ((t >> 3) + i * (NT / 8)) * LDK + (t & 7) * 4) = v[i];
            }
        } else {
            constexpr int CPR = R / 4;
            *reinterpret_cast<float4*>(lds + (t / CPR + i * (NT / CPR)) * R + (t % CPR) * 4) = v[i];
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds) const {
        const int t = threadIdx.x;
        if (KC) {
            const int c = (t & 7) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = (t >> 3) + i * (NT / 8);
                if (KCT) {
                    lds[(c + 0) * (R + 1) + r] = v[i].x; lds[(c + 1) * (R + 1) + r] = v[i].y;
                    lds[(c + 2) * (R + 1) + r] = v[i].z; lds[(c + 3) * (R + 1) + r] = v[i].w;
                } else {
                    *reinterpret_cast<float4*>(lds + r * LDK + c) = v[i];
                }
            }
        } else {
            constexpr int CPR = R / 4;
            const int c = (t % CPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = t / CPR + i * (NT / CPR);
                *reinterpret_cast<float4*>(lds + k * R + c) = v[i];
            }
        }
    }
};

template <int R, bool KC>
constexpr int stage_floats() { return KC ? (KCT ? ((BK * (R + 1) + 3) / 4) * 4 : R * LDK) : BK * R; }

// Reads this lane's 8 k-values (k = 16*h + 8*half + j) of operand row `row` from the LDS image.
template <int R, bool KC>
__device__ __forceinline__ void read_frag(const float* __restrict__ lds, int row, int h, int half, float (&f)[8]) {
    if (KC && KCT) {
        const float* p = lds + (16 * h + 8 * half) * (R + 1) + row;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = p[j * (R + 1)];
    } else if (KC) {
        const float4* p = reinterpret_cast<const float4*>(lds + row * LDK + 16 * h + 8 * half);
        const float4 a = p[0], b = p[1];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
        f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        const float* p = lds + (16 * h + 8 * half) * R + row;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = p[j * R];
    }
}

template <int BM, int BN, int WGM, int WGN, int PIPE_ABL, bool A_KC, bool B_KC, bool VA, bool VB>
__global__ void __launch_bounds__(WGM* WGN * 64) gemm_f32_kernel(const GemmParams p_in) {
    GemmParams p = p_in;
    p.dk = drop_key_now(p.dk);          // graph replays: seed + device offset (lstc_dropout_seed_device)
    p.A += (size_t)blockIdx.z * p.batch_stride_a;
    p.B += (size_t)blockIdx.z * p.batch_stride_b;
    p.C += (size_t)blockIdx.z * p.batch_stride_c;
    constexpr int PIPE = PIPE_ABL & 15;
    constexpr int ABL = PIPE_ABL >> 4;     // timing-only ablation of the PIPE 3 body: 1 no global loads, 2 no LDS writes, 4 no barrier
    constexpr int NT = WGM * WGN * 64;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_ST = stage_floats<BM, A_KC>();
    constexpr int B_ST = stage_floats<BN, B_KC>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                  // 2 stages
    float* const Bs = smem + 2 * A_ST;       // 2 stages

    // ---- XCD-aware, bijective workgroup -> tile map
    const int nwg = p.tilesM * p.tilesN;
    int pid = blockIdx.x;
    {
        const int xcd = pid & 7, idx = pid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
#if LSTC_F32_GROUP_M
    // grouped order: inside the linear tile index, groups of GROUP_M consecutive M panels are walked M fastest, so the 64 tiles
    // resident on an XCD cover 8 panels x 8 N tiles instead of 4 x 16 (N = 2048): per K step they touch 8 + 8 operand chunks
    // instead of 4 + 16.  Same-box A/B (tools/f32_group_ab.sh, round 4), 100352 x 2048 x 2048: FETCH_SIZE 3.61e6 -> 2.41e6 KB (HBM-side
    // traffic per launch 8.04 -> 5.64 GB), L2 hit rate 70.6 -> 79.4 %, 5.62 -> 5.60 ms (NT), 5.65 -> 5.61 (NN), N = 4096 11.27 -> 11.21
    int mt, nt;
    {
        constexpr int GM = LSTC_F32_GROUP_M;
        const int per_group = GM * p.tilesN;
        const int gid = pid / per_group, first_m = gid * GM;
        const int gsz = min(p.tilesM - first_m, GM);
        const int loc = pid - gid * per_group;
        mt = first_m + loc % gsz;
        nt = loc / gsz;
    }
#else
    const int mt = pid / p.tilesN, nt = pid % p.tilesN;
#endif
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt0 = blockIdx.y * p.ktiles_per_split;
    const int kt1 = min(p.ktiles, kt0 + p.ktiles_per_split);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, h = lane >> 5;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stager<BM, NT, A_KC, VA> sa;
    Stager<BN, NT, B_KC, VB> sb;
    const int nkt = kt1 - kt0;
    const bool k_tail = (p.K % BK) != 0;          // only the globally last K tile can be partial

    auto gload = [&](int kt) {
        if (k_tail && kt == p.ktiles - 1) {
            sa.template load<true>(p.A, p.lda, m0, p.M, kt * BK, p.K);
            sb.template load<true>(p.B, p.ldb, n0, p.N, kt * BK, p.K);
        } else {
            sa.template load<false>(p.A, p.lda, m0, p.M, kt * BK, p.K);
            sb.template load<false>(p.B, p.ldb, n0, p.N, kt * BK, p.K);
        }
    };

    // Software pipeline with ONE barrier per K tile, placed in the middle of the tile's MFMAs:
    //   phase 1: LDS-write tile t+1 (registers, loaded one iteration ago) | issue global loads of tile t+2 |
    //            LDS-read the 2nd-half fragments of tile t      -> all hidden behind the 1st-half MFMAs of tile t
    //   barrier: publishes tile t+1 and retires every read of tile t's stage
    //   phase 2: LDS-read the 1st-half fragments of tile t+1   -> hidden behind the 2nd-half MFMAs of tile t
    // so a wave reaches each MFMA block with its operands already in registers; only barrier skew is exposed.
    if constexpr (PIPE == 4) {
        // ---- LDS-DMA staging (global_load_lds_dwordx4): operand tiles go HBM/L2 -> LDS without passing through
        // VGPRs, so the steady state has no ds_write (2.6 % of the PIPE 3 time) and no staging registers.
        // A DMA wave-instruction writes 1 KB lane-linearly (base + 16*lane), so the LDS images are UNPADDED:
        //   K-contiguous operand: [rows][32 floats]; bank conflicts of the ds_read_b128 fragment reads are removed
        //     by XOR-ing the 16-B chunk index with (row>>1)&7 — applied to the per-lane SOURCE address of the DMA and
        //     to the read address (cdna_hip_programming rule 21: same involution on both sides, linear destination);
        //   k-major operand: [32][rows], naturally lane-linear (one piece = 2 k rows of 128 floats).
        // Pipeline: tile t+2 is DMA'd into the stage tile t just vacated, right after the mid-tile barrier of
        // iteration t; __syncthreads() of iteration t+1 (which carries s_waitcnt vmcnt(0)) publishes it.
        static_assert(VA && VB, "LDS-DMA path needs 16-B aligned rows");
        constexpr int ST = 32 * BM, STB = 32 * BN;          // floats per stage
        float* const A4 = smem;
        float* const B4 = smem + 2 * ST;
        constexpr int PA = BM / 32, PB = BN / 32;           // 1-KB pieces per wave per stage (NT = 256: 4 waves)
        static_assert(NT == 256, "piece assignment assumes 4 waves");
        // per-lane source pointers of this wave's pieces at k = 0 (rows clamped; k advances by pointer arithmetic)
        const float* ga[PA];
        const float* gb[PB];
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int pi = wave * PA + j;
            if (A_KC) {
                const int row = pi * 8 + (lane >> 3), cp = lane & 7;
                ga[j] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + ((cp ^ ((row >> 1) & 7)) << 2);
            } else {
                const int f = pi * 256 + lane * 4;
                ga[j] = p.A + (size_t)(f / BM) * p.lda + min(m0 + f % BM, p.M - 4);
            }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int pi = wave * PB + j;
            if (B_KC) {
                const int row = pi * 8 + (lane >> 3), cp = lane & 7;
                gb[j] = p.B + (size_t)min(n0 + row, p.N - 1) * p.ldb + ((cp ^ ((row >> 1) & 7)) << 2);
            } else {
                const int f = pi * 256 + lane * 4;
                gb[j] = p.B + (size_t)(f / BN) * p.ldb + min(n0 + f % BN, p.N - 4);
            }
        }
        const size_t a_kstep = A_KC ? (size_t)BK : (size_t)BK * p.lda;     // floats per K tile
        const size_t b_kstep = B_KC ? (size_t)BK : (size_t)BK * p.ldb;
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        auto dma_a = [&](int j, int kt, int stage) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ga[j] + (size_t)kt * a_kstep),
                                             (lptr_t)(A4 + stage * ST + (wave * PA + j) * 256), 16, 0, 0);
        };
        auto dma_b = [&](int j, int kt, int stage) {
            __builtin_amdgcn_global_load_lds((gptr_t)(gb[j] + (size_t)kt * b_kstep),
                                             (lptr_t)(B4 + stage * STB + (wave * PB + j) * 256), 16, 0, 0);
        };
        // fragment read from the unpadded images
        auto frag = [&](const float* lds, bool kc, int R, int row, int half, float (&f)[8]) {
            if (kc) {
                const int c0 = 4 * h + 2 * half, sw = (row >> 1) & 7;
                const float4 a = *reinterpret_cast<const float4*>(lds + row * 32 + ((c0 ^ sw) << 2));
                const float4 b = *reinterpret_cast<const float4*>(lds + row * 32 + (((c0 + 1) ^ sw) << 2));
                f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
            } else {
                const float* q = lds + (16 * h + 8 * half) * R + row;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = q[j * R];
            }
        };
        float fa0[TM][8], fb0[TN][8], fa1[TM][8], fb1[TN][8];
        if (nkt > 0) {
#pragma unroll
            for (int j = 0; j < PA; ++j) dma_a(j, kt0, 0);
#pragma unroll
            for (int j = 0; j < PB; ++j) dma_b(j, kt0, 0);
            if (nkt > 1) {
#pragma unroll
                for (int j = 0; j < PA; ++j) dma_a(j, kt0 + 1, 1);
#pragma unroll
                for (int j = 0; j < PB; ++j) dma_b(j, kt0 + 1, 1);
            }
        }
        __syncthreads();
        if (nkt > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) frag(A4, A_KC, BM, wm * WTM + i * 32 + l31, 0, fa0[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) frag(B4, B_KC, BN, wn * WTN + j * 32 + l31, 0, fb0[j]);
        }
        for (int it = 0; it < nkt; ++it) {
            const int cur = it & 1;
            const float* a_lds = A4 + cur * ST;
            const float* b_lds = B4 + cur * STB;
            const float* a_nx = A4 + (cur ^ 1) * ST;
            const float* b_nx = B4 + (cur ^ 1) * STB;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {           // phase 1: 2nd-half fragments of tile t behind its 1st-half MFMAs
                if (kk < TM) frag(a_lds, A_KC, BM, wm * WTM + kk * 32 + l31, 1, fa1[kk < TM ? kk : 0]);
                else if (kk - TM < TN) frag(b_lds, B_KC, BN, wn * WTN + (kk - TM) * 32 + l31, 1, fb1[kk - TM < TN ? kk - TM : 0]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[i][kk], fb0[j][kk], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                           // tile t+1 landed (vmcnt(0)) and every read of tile t retired
            const bool more1 = it + 1 < nkt, more2 = it + 2 < nkt;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {           // phase 2: DMA tile t+2 into the freed stage; frags of tile t+1
                if (more2) {
                    if (kk < PA) dma_a(kk < PA ? kk : 0, kt0 + it + 2, cur);
                    else if (kk - PA < PB) dma_b(kk - PA < PB ? kk - PA : 0, kt0 + it + 2, cur);
                }
                if (more1) {
                    if (kk < TM) frag(a_nx, A_KC, BM, wm * WTM + kk * 32 + l31, 0, fa0[kk < TM ? kk : 0]);
                    else if (kk - TM < TN) frag(b_nx, B_KC, BN, wn * WTN + (kk - TM) * 32 + l31, 0, fb0[kk - TM < TN ? kk - TM : 0]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[i][kk], fb1[j][kk], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    if constexpr (PIPE == 0) {
        // Plain double buffering: global loads of tile t+1 issued before the MFMAs of tile t, LDS write after them,
        // one barrier at the end of the tile (fragment reads of the next tile are exposed after the barrier).
        if (nkt > 0) {
            gload(kt0);
            sa.store(As);
            sb.store(Bs);
        }
        __syncthreads();
        for (int it = 0; it < nkt; ++it) {
            const int cur = it & 1;
            const bool more = it + 1 < nkt;
            if (more && !(p.debug & 1)) gload(kt0 + it + 1);
            const float* a_lds = As + cur * A_ST;
            const float* b_lds = Bs + cur * B_ST;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float af[TM][8], bf[TN][8];
#pragma unroll
                for (int i = 0; i < TM; ++i) read_frag<BM, A_KC>(a_lds, wm * WTM + i * 32 + l31, h, half, af[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) read_frag<BN, B_KC>(b_lds, wn * WTN + j * 32 + l31, h, half, bf[j]);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
            }
            if (more && !(p.debug & 2)) {
                sa.store(As + (cur ^ 1) * A_ST);
                sb.store(Bs + (cur ^ 1) * B_ST);
            }
            if (!(p.debug & 4)) __syncthreads();
        }
    } else {
    float fa0[TM][8], fb0[TN][8], fa1[TM][8], fb1[TN][8];
    if (nkt > 0) {
        gload(kt0);
        sa.store(As);
        sb.store(Bs);
        if (nkt > 1) gload(kt0 + 1);
    }
    __syncthreads();
    if (nkt > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) read_frag<BM, A_KC>(As, wm * WTM + i * 32 + l31, h, 0, fa0[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) read_frag<BN, B_KC>(Bs, wn * WTN + j * 32 + l31, h, 0, fb0[j]);
    }
    // One K tile.  STEADY (compile-time): tiles it+1 and it+2 exist and are full -> branch-free body whose memory
    // instructions are interleaved with the MFMAs by sched_group_barrier (PIPE == 2): a wave-level global load that
    // touches 8 separate 128-B lines holds the issue port for tens of cycles; clustered at the top of the tile they
    // delayed the MFMA stream of BOTH waves of the SIMD (measured: -11 % on the NT layout), spread two MFMAs apart
    // they disappear in the 64-cycle shadow of each MFMA.
    // PIPE 5 = PIPE 3 with fewer non-MFMA instructions in the steady loop (tools/mfma_issue_probe: every companion
    // instruction costs the MFMA pipe ~7 idle cycles even at two waves per SIMD; PIPE 3 carried 72 per 64 MFMAs):
    // global loads become buffer_load_dwordx4 with a loop-invariant per-lane voffset and a scalar K offset (no 64-bit
    // vector address arithmetic), and the K loop is unrolled by two so LDS stage bases are immediates.
    constexpr int NVA5 = Stager<BM, NT, A_KC, VA>::NV, NVB5 = Stager<BN, NT, B_KC, VB>::NV;
    uint32_t voff_a[NVA5], voff_b[NVB5];
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
    if constexpr (PIPE == 5) {
#pragma unroll
        for (int e = 0; e < NVA5; ++e) voff_a[e] = sa.voffset(e, p.lda, m0, p.M);
#pragma unroll
        for (int e = 0; e < NVB5; ++e) voff_b[e] = sb.voffset(e, p.ldb, n0, p.N);
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
        rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, (int)p.b_bytes, 0x00020000);
    }
    const uint32_t a_kbytes = (uint32_t)((A_KC ? (size_t)BK : (size_t)BK * p.lda) * sizeof(float));
    const uint32_t b_kbytes = (uint32_t)((B_KC ? (size_t)BK : (size_t)BK * p.ldb) * sizeof(float));

    // steady_tag: 0 = generic body (runtime conditions, checked loads); 1 = tiles t+1 and t+2 exist; 2 = only t+1 exists
    // (no more loads); 3 = last tile.  Modes 2/3 keep the hand-interleaved form for the tail of the K loop (PIPE 5).
    auto tile_step = [&](int it, auto steady_tag, auto cur_tag) {
        constexpr int SMODE = decltype(steady_tag)::value;
        constexpr bool STEADY = SMODE != 0;
        constexpr bool HAS1 = SMODE == 1 || SMODE == 2, HAS2 = SMODE == 1;
        constexpr int CC = decltype(cur_tag)::value;
        const int cur = CC >= 0 ? CC : (it & 1);
        const float* a_lds = As + cur * A_ST;
        const float* b_lds = Bs + cur * B_ST;
        if constexpr (STEADY && (PIPE == 3 || PIPE == 5)) {
            // Hand-interleaved steady state: each group of TM*TN independent MFMAs (one k step, all accumulators)
            // carries at most one LDS write, one global load and one fragment read, pinned by sched_barrier so the
            // compiler neither clusters the memory instructions nor chains MFMAs on one accumulator.
            constexpr int NVA = Stager<BM, NT, A_KC, VA>::NV, NVB = Stager<BN, NT, B_KC, VB>::NV;
            constexpr int MPK = (NVA + NVB + 7) / 8;          // staged float4 per k step (1 for 128x128, 2 for 256x256)
            static_assert(TM + TN <= 8, "one fragment read per k step");
            float* a_st = As + (cur ^ 1) * A_ST;
            float* b_st = Bs + (cur ^ 1) * B_ST;
            const int k_next = (kt0 + it + 2) * BK;
            const uint32_t soff_a = (uint32_t)(kt0 + it + 2) * a_kbytes, soff_b = (uint32_t)(kt0 + it + 2) * b_kbytes;
            if constexpr (PIPE == 5 && HAS1) {
                // the staged loads were issued 12+ k steps ago: ONE wait for all of them instead of one per LDS write
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), lgkmcnt / expcnt untouched
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                // k steps 0-3: LDS-write the staged tile t+1 (loaded one iteration ago); k steps 4-7: issue the global
                // loads of tile t+2 into the same registers.  Keeping every write ahead of every new load matters:
                // vmcnt retires in order, and with the two interleaved the compiler's wait before a write also
                // forced the newest loads of THIS iteration to land (s_waitcnt vmcnt(5): ~10 % stall).
#pragma unroll
                for (int q = 0; q < 2 * MPK; ++q) {
                    const int e = (kk & 3) * 2 * MPK + q;
                    if (kk < 4) {
                        if constexpr (!(ABL & 2) && HAS1) {
                            if (e < NVA) sa.store_one(e, a_st);
                            else if (e - NVA < NVB) sb.store_one(e - NVA, b_st);
                        }
                    } else if constexpr (HAS2) {
                        if constexpr (PIPE == 5) {
                            if (e < NVA) sa.load_buf(e, rsrc_a, voff_a[e < NVA ? e : 0], soff_a);
                            else if (e - NVA < NVB) sb.load_buf(e - NVA, rsrc_b, voff_b[e - NVA < NVB ? e - NVA : 0], soff_b);
                        } else if constexpr (!(ABL & 1)) {
                            if (e < NVA) sa.load_one(e, p.A, p.lda, m0, p.M, k_next);
                            else if (e - NVA < NVB) sb.load_one(e - NVA, p.B, p.ldb, n0, p.N, k_next);
                        }
                    }
                }
                if (kk < TM) read_frag<BM, A_KC>(a_lds, wm * WTM + kk * 32 + l31, h, 1, fa1[kk < TM ? kk : 0]);
                else if (kk - TM < TN) read_frag<BN, B_KC>(b_lds, wn * WTN + (kk - TM) * 32 + l31, h, 1, fb1[kk - TM < TN ? kk - TM : 0]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[i][kk], fb0[j][kk], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!(ABL & 4)) __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if constexpr (HAS1) {
                    if (kk < TM) read_frag<BM, A_KC>(a_st, wm * WTM + kk * 32 + l31, h, 0, fa0[kk < TM ? kk : 0]);
                    else if (kk - TM < TN) read_frag<BN, B_KC>(b_st, wn * WTN + (kk - TM) * 32 + l31, h, 0, fb0[kk - TM < TN ? kk - TM : 0]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[i][kk], fb1[j][kk], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        if (STEADY || it + 1 < nkt) {
            sa.store(As + (cur ^ 1) * A_ST);
            sb.store(Bs + (cur ^ 1) * B_ST);
        }
        if (STEADY) {
            sa.template load<false>(p.A, p.lda, m0, p.M, (kt0 + it + 2) * BK, p.K);
            sb.template load<false>(p.B, p.ldb, n0, p.N, (kt0 + it + 2) * BK, p.K);
        } else if (it + 2 < nkt) {
            gload(kt0 + it + 2);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) read_frag<BM, A_KC>(a_lds, wm * WTM + i * 32 + l31, h, 1, fa1[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) read_frag<BN, B_KC>(b_lds, wn * WTN + j * 32 + l31, h, 1, fb1[j]);
        if (!(STEADY && PIPE == 2)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[i][kk], fb0[j][kk], acc[i][j], 0, 0, 0);
        if constexpr (STEADY && PIPE == 2 && VA && VB) {
            constexpr int NRD = (A_KC ? 2 * TM : 8 * TM) + (B_KC ? 2 * TN : 8 * TN);     // fragment reads
            constexpr int NWR = Stager<BM, NT, A_KC, VA>::NV + Stager<BN, NT, B_KC, VB>::NV;   // ds_write_b128 == global loads
            constexpr int NMF = TM * TN * 8;
            constexpr int PER = NMF / (2 * NWR) > 0 ? NMF / (2 * NWR) : 1;
            __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);              // fragment reads first (needed after the barrier)
#pragma unroll
            for (int q = 0; q < NWR; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);            // ds_write  (tile it+1)
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);          // MFMA
            }
#pragma unroll
            for (int q = 0; q < NWR; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            // global load (tile it+2)
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (STEADY || it + 1 < nkt) {
            const float* a_nx = As + (cur ^ 1) * A_ST;
            const float* b_nx = Bs + (cur ^ 1) * B_ST;
#pragma unroll
            for (int i = 0; i < TM; ++i) read_frag<BM, A_KC>(a_nx, wm * WTM + i * 32 + l31, h, 0, fa0[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) read_frag<BN, B_KC>(b_nx, wn * WTN + j * 32 + l31, h, 0, fb0[j]);
        }
        if (!(STEADY && PIPE == 2)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[i][kk], fb1[j][kk], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int it = 0;
    using dyn = std::integral_constant<int, -1>;
    using gen = std::integral_constant<int, 0>;
    using full = std::integral_constant<int, 1>;
    if constexpr (PIPE == 5) {
        // unchecked loads may fetch tile it+2 only if it is a full K tile: when this workgroup owns the final, partial
        // K tile the interleaved form stops one iteration earlier and the generic (checked) body finishes the loop
        const bool tail_ok = !(k_tail && kt1 == p.ktiles);
        const int full_end = tail_ok ? nkt - 2 : nkt - 3;
        for (; it + 1 < full_end; it += 2) {
            tile_step(it, full{}, std::integral_constant<int, 0>{});
            tile_step(it + 1, full{}, std::integral_constant<int, 1>{});
        }
        for (; it < full_end; ++it) tile_step(it, full{}, dyn{});
        if (tail_ok) {
            if (it + 1 < nkt) { tile_step(it, std::integral_constant<int, 2>{}, dyn{}); ++it; }
            if (it < nkt) { tile_step(it, std::integral_constant<int, 3>{}, dyn{}); ++it; }
        }
    }
    for (; it + 3 < nkt; ++it) tile_step(it, full{}, dyn{});
    for (; it < nkt; ++it) tile_step(it, gen{}, dyn{});
    }

    // ---- epilogue
    if constexpr (TM <= 2 && TN <= 2 && (PIPE_ABL & 15) == 5) {
        if (p.epi_f4 && gridDim.y == 1) {                       // float4 form (host checked alignment and the operand count)
            const EpiArgs ea = LSTC_EPI_ARGS(p);
            if (p.epi_f4 == 2) epilogue_f4<TM, TN, true>(ea, acc, m0 + wm * WTM, n0 + wn * WTN, lane);
            else epilogue_f4<TM, TN, false>(ea, acc, m0 + wm * WTM, n0 + wn * WTN, lane);
            return;
        }
    }
    const int flags = p.flags;
    const bool atomic = gridDim.y > 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = (flags & LSTC_EPI_BIAS) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + wm * WTM + i * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row >= p.M) continue;
                float v = acc[i][j][r] * p.alpha;
                float* cp = p.C + (size_t)row * p.ldc + col;
                if (atomic) {
                    atomicAdd(cp, v);
                    continue;
                }
                v += bv;
                if (flags & LSTC_EPI_RELU) v = fmaxf(v, 0.f);
                if (flags & LSTC_EPI_DROPOUT) {
                    const uint32_t idx = (uint32_t)(row + p.row_off) * (uint32_t)p.N + (uint32_t)col;
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

// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the PIPE 5 kernel (variant 12; aligned operands, K a multiple of 32, >= 4 K tiles, no split-K, no batch).
// Per-tile fixed cost of the one-tile-per-workgroup kernel: t(tile) = 12.3 us + 0.1087 us/k on the headline shapes (K = 2048:
// 234.9 us, K = 4096: 457.5 us) - first-load latency of the prologue plus 64 scattered 4-byte stores per lane in the epilogue,
// 5 % of a K = 2048 tile.  Here min(tiles, 2 x CUs) workgroups walk the tiles round by round (per round an XCD's workgroups own
// one contiguous run of tiles, N fastest), and
//   * the first K tile of the NEXT output tile is requested into the staging registers BEFORE the epilogue of the current one
//     (its latency runs under the epilogue's stores),
//   * the epilogue transposes each 4-register group across its lane quad (two DPP quad_perm steps), so a lane owns four
//     consecutive columns of one row: 16 global_store_dwordx4 per lane instead of 64 global_store_dword, every wave-level
//     store = 8 rows x 128-B full lines, and bias / residual / ReLU-mask operands are float4 loads.
// The K loop and the k order of every output element are those of PIPE 5: results are bit-identical to variant 4.
template <bool A_KC, bool B_KC, bool AUX>
__global__ void __launch_bounds__(256) gemm_f32_persist_kernel(const GemmParams p_in) {
    GemmParams p = p_in;
    p.dk = drop_key_now(p.dk);
    constexpr int BM = 128, BN = 128, WGN = 2, NT = 256, WTM = 64, WTN = 64, TM = 2, TN = 2;
    constexpr int A_ST = stage_floats<BM, A_KC>();
    constexpr int B_ST = stage_floats<BN, B_KC>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;
    float* const Bs = smem + 2 * A_ST;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, h = lane >> 5;
    const int ntiles = p.tilesM * p.tilesN;
    const int G = gridDim.x, per = G >> 3;                  // G is a multiple of 8 (launcher)
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int nkt = p.ktiles;                               // >= 4, all full (launcher)
    // tile of round r for this workgroup (-1: none).  Full rounds: XCD x owns tiles [r G + x per, r G + (x + 1) per).  The last,
    // partial round deals its tiles to the XCDs evenly (first `pl` workgroups of every XCD) instead of filling XCD 0 first.
    auto tile_of = [&](int r) -> int {
        const int base = r * G, rem = ntiles - base;
        if (rem <= 0) return -1;
        if (rem >= G) return base + xcd * per + loc;
        const int pl = (rem + 7) >> 3;
        const int t = base + xcd * pl + loc;
        return (loc < pl && xcd * pl + loc < rem) ? t : -1;
    };
    constexpr int NVA = Stager<BM, NT, A_KC, true>::NV, NVB = Stager<BN, NT, B_KC, true>::NV;
    static_assert(NVA + NVB <= 8, "one staged float4 per k step and slot");
    Stager<BM, NT, A_KC, true> sa;
    Stager<BN, NT, B_KC, true> sb;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, (int)p.b_bytes, 0x00020000);
    const uint32_t a_kbytes = (uint32_t)((A_KC ? (size_t)BK : (size_t)BK * p.lda) * sizeof(float));
    const uint32_t b_kbytes = (uint32_t)((B_KC ? (size_t)BK : (size_t)BK * p.ldb) * sizeof(float));
    uint32_t voff_a[NVA], voff_b[NVB];

    int round = 0;
    int tile = tile_of(0);
    bool pre = false;               // the first K tile of `tile` is already in the staging registers
    while (tile >= 0) {
        const int mt = tile / p.tilesN, nt = tile - mt * p.tilesN;
        const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
        for (int e = 0; e < NVA; ++e) voff_a[e] = sa.voffset(e, p.lda, m0, p.M);
#pragma unroll
        for (int e = 0; e < NVB; ++e) voff_b[e] = sb.voffset(e, p.ldb, n0, p.N);
        floatx16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (!pre) {
#pragma unroll
            for (int e = 0; e < NVA; ++e) sa.load_buf(e, rsrc_a, voff_a[e], 0u);
#pragma unroll
            for (int e = 0; e < NVB; ++e) sb.load_buf(e, rsrc_b, voff_b[e], 0u);
        }
        __syncthreads();                       // every LDS read of the previous tile is complete
        sa.store(As);
        sb.store(Bs);
#pragma unroll
        for (int e = 0; e < NVA; ++e) sa.load_buf(e, rsrc_a, voff_a[e], a_kbytes);
#pragma unroll
        for (int e = 0; e < NVB; ++e) sb.load_buf(e, rsrc_b, voff_b[e], b_kbytes);
        __syncthreads();
        float fa0[TM][8], fb0[TN][8], fa1[TM][8], fb1[TN][8];
#pragma unroll
        for (int i = 0; i < TM; ++i) read_frag<BM, A_KC>(As, wm * WTM + i * 32 + l31, h, 0, fa0[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) read_frag<BN, B_KC>(Bs, wn * WTN + j * 32 + l31, h, 0, fb0[j]);

        // one K tile: the hand-interleaved PIPE 5 body (SMODE 1: tiles it+1, it+2 exist; 2: only it+1; 3: last tile)
        auto tile_step = [&](int it, auto steady_tag, auto cur_tag) {
            constexpr int SMODE = decltype(steady_tag)::value;
            constexpr bool HAS1 = SMODE == 1 || SMODE == 2, HAS2 = SMODE == 1;
            constexpr int CC = decltype(cur_tag)::value;
            const int cur = CC >= 0 ? CC : (it & 1);
            const float* a_lds = As + cur * A_ST;
            const float* b_lds = Bs + cur * B_ST;
            float* a_st = As + (cur ^ 1) * A_ST;
            float* b_st = Bs + (cur ^ 1) * B_ST;
            const uint32_t soff_a = (uint32_t)(it + 2) * a_kbytes, soff_b = (uint32_t)(it + 2) * b_kbytes;
            if constexpr (HAS1) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the staged tile it+1 has landed
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = (kk & 3) * 2 + q;
                    if (kk < 4) {
                        if constexpr (HAS1) {
                            if (e < NVA) sa.store_one(e, a_st);
                            else if (e - NVA < NVB) sb.store_one(e - NVA, b_st);
                        }
                    } else if constexpr (HAS2) {
                        if (e < NVA) sa.load_buf(e, rsrc_a, voff_a[e < NVA ? e : 0], soff_a);
                        else if (e - NVA < NVB) sb.load_buf(e - NVA, rsrc_b, voff_b[e - NVA < NVB ? e - NVA : 0], soff_b);
                    }
                }
                if (kk < TM) read_frag<BM, A_KC>(a_lds, wm * WTM + kk * 32 + l31, h, 1, fa1[kk < TM ? kk : 0]);
                else if (kk - TM < TN) read_frag<BN, B_KC>(b_lds, wn * WTN + (kk - TM) * 32 + l31, h, 1, fb1[kk - TM < TN ? kk - TM : 0]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[i][kk], fb0[j][kk], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if constexpr (HAS1) {
                    if (kk < TM) read_frag<BM, A_KC>(a_st, wm * WTM + kk * 32 + l31, h, 0, fa0[kk < TM ? kk : 0]);
                    else if (kk - TM < TN) read_frag<BN, B_KC>(b_st, wn * WTN + (kk - TM) * 32 + l31, h, 0, fb0[kk - TM < TN ? kk - TM : 0]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[i][kk], fb1[j][kk], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using dyn = std::integral_constant<int, -1>;
        using full = std::integral_constant<int, 1>;
        int it = 0;
        const int full_end = nkt - 2;
        for (; it + 1 < full_end; it += 2) {
            tile_step(it, full{}, std::integral_constant<int, 0>{});
            tile_step(it + 1, full{}, std::integral_constant<int, 1>{});
        }
        for (; it < full_end; ++it) tile_step(it, full{}, dyn{});
        tile_step(it, std::integral_constant<int, 2>{}, dyn{}); ++it;
        tile_step(it, std::integral_constant<int, 3>{}, dyn{});

        // ---- the next tile's first K tile is requested before this tile's epilogue
        ++round;
        const int next = tile_of(round);
        pre = false;
        if (next >= 0) {
            const int mtn = next / p.tilesN, ntn = next - mtn * p.tilesN;
#pragma unroll
            for (int e = 0; e < NVA; ++e) sa.load_buf(e, rsrc_a, sa.voffset(e, p.lda, mtn * BM, p.M), 0u);
#pragma unroll
            for (int e = 0; e < NVB; ++e) sb.load_buf(e, rsrc_b, sb.voffset(e, p.ldb, ntn * BN, p.N), 0u);
            pre = true;
        }
        __builtin_amdgcn_sched_barrier(0);

        epilogue_f4<TM, TN, AUX>(LSTC_EPI_ARGS(p), acc, m0 + wm * WTM, n0 + wn * WTN, lane);
        tile = next;
    }
}

template <bool A_KC, bool B_KC>
int launch_persist(const GemmParams& p, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * stage_floats<128, A_KC>() + 2 * stage_floats<128, B_KC>()) * sizeof(float);
    static std::atomic<int> slots_dev[64];
    static LstcDevOnce once;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    const int dev_ = once.begin();
    if (dev_ >= 0) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev_) == hipSuccess) n = prop.multiProcessorCount;
        slots_dev[dev_ & 63].store(n > 0 ? 2 * n : 512, std::memory_order_relaxed);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_persist_kernel<A_KC, B_KC, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_persist_kernel<A_KC, B_KC, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once.end(dev_);
    }
    int G = slots_dev[cur & 63].load(std::memory_order_relaxed);
    const int ntiles = p.tilesM * p.tilesN;
    if (G > ntiles) G = ntiles;
    G &= ~7;
    if (G < 8) return LSTC_E_UNSUPPORTED;
    // per-element operand of the epilogue: at most one of residual / ReLU-mask source / accumulate target on this kernel
    if (p.epi_f4 == 2) hipLaunchKernelGGL((gemm_f32_persist_kernel<A_KC, B_KC, true>), dim3(G), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemm_f32_persist_kernel<A_KC, B_KC, false>), dim3(G), dim3(256), lds, st, p);
    return lstc_launch_status();
}

template <int BM, int BN, int WGM, int WGN, int PIPE, bool A_KC, bool B_KC>
int launch_cfg(const GemmParams& p, bool va, bool vb, int splits, hipStream_t st) {
    constexpr int NT = WGM * WGN * 64;
    constexpr size_t lds = (size_t)(2 * stage_floats<BM, A_KC>() + 2 * stage_floats<BN, B_KC>()) * sizeof(float);
    dim3 grid(p.tilesM * p.tilesN, splits, p.batch), block(NT);
#define LSTC_GO(VA, VB)                                                                                     \
    do {                                                                                                    \
        auto kern = gemm_f32_kernel<BM, BN, WGM, WGN, PIPE, A_KC, B_KC, VA, VB>;                                  \
        static LstcDevOnce attr_done;                                                                       \
        const int dev_ = attr_done.begin();                                                                 \
        if (dev_ >= 0) {                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                  \
            attr_done.end(dev_);                                                                            \
        }                                                                                                   \
        hipLaunchKernelGGL(kern, grid, block, lds, st, p);                                                  \
    } while (0)
    if constexpr ((PIPE & 15) == 4 || (PIPE & 15) == 5) {
        LSTC_GO(true, true);               // LDS-DMA / buffer-load paths: aligned operands only (caller guarantees va && vb)
    } else {
        if (va && vb) LSTC_GO(true, true);
        else if (va) LSTC_GO(true, false);
        else if (vb) LSTC_GO(false, true);
        else LSTC_GO(false, false);
    }
#undef LSTC_GO
    return lstc_launch_status();
}

template <bool A_KC, bool B_KC>
int launch_layout(GemmParams& p, bool va, bool vb, int splits, int variant, hipStream_t st) {
    variant &= 15;
    // variant: 0 = library default.  Measured on MI355X, LTN shapes, TFLOP/s NT / NN / TN(split-K 4)
    // (profiles/r01_gemm_variants.log, tools/gemm_check):
    //   1 = 128x128, plain double buffering (PIPE 0)                              117 / 129 / 134
    //   7 = 128x128, mid-barrier software pipeline (PIPE 1)                       126 / 127 / 128
    //   3 = 128x128, pipeline + sched_group_barrier interleave (PIPE 2)           136 / 131 / 129
    //   8 = 128x128, pipeline + hand-interleaved k-step groups (PIPE 3)           138 / 136 / 138   (fallback of 4)
    //   4 = PIPE 3 + buffer loads with scalar K offset, K loop unrolled x2 (PIPE 5) 145 / 143 / 151   <- default
    //       (steady loop: 49 instead of 72 non-MFMA instructions per 64 MFMAs; needs 16-B aligned operands < 4 GiB)
    //   9 = 256x128, 4 waves x (128x64), one wave per SIMD, PIPE 3                125 / 124 / 134
    //  11 = 64x64, 4 waves x (32x32), plain double buffering: same k order per output element as every other variant, used for
    //       the last rows of a product whose 128x128 tile count leaves the final round of workgroup slots mostly empty
    //  10 = 128x128, LDS-DMA staging (global_load_lds, swizzled unpadded images)  105 / 119 / 134   (correct, slower:
    //       the swizzled per-lane source addresses of K-contiguous operands and the one-iteration latency budget cost
    //       more than the ds_write + staging registers they remove)
    //   2, 6, 5 = 256x128 with 8 waves (PIPE 1 / 2 / 0)                           115-126, never the best
    //  13-15 = (-DLSTC_TUNING builds only) timing-only ablations of variant 8 (NT): no loads 142, no loads/LDS writes 146,
    //          +no barrier 146.5.  12 is the persistent kernel in EVERY build.
    // The PRODUCTION library (`make all`) holds 0 = 4, 8 (the default's fallback), 11 and 12; variants 1-3, 5-7, 9, 10 are compiled into
    // `make tuning` builds only (tools/tuning/liblstc_hip.so) and refused here otherwise.
    if (variant == 0) variant = 4;
    const int BM = (variant == 2 || variant == 5 || variant == 6 || variant == 9) ? 256 : variant == 11 ? 64 : 128, BN = variant == 11 ? 64 : 128;
    p.tilesM = (p.M + BM - 1) / BM;
    p.tilesN = (p.N + BN - 1) / BN;
    switch (variant) {
        case 11: return launch_cfg<64, 64, 2, 2, 0, A_KC, B_KC>(p, va, vb, splits, st);    // small tile: the tail rows of a row-split product
        //  12 = persistent PIPE 5 (gemm_f32_persist_kernel): next tile's first loads before the epilogue, float4 epilogue
        case 12: if (va && vb && p.a_bytes && p.b_bytes && splits == 1 && p.batch == 1 && p.K % BK == 0 && p.ktiles >= 4 && p.epi_f4)
                 {
                     const int rc_ = launch_persist<A_KC, B_KC>(p, st);
                     if (rc_ != LSTC_E_UNSUPPORTED) return rc_;
                 }
                 if (va && vb && p.a_bytes && p.b_bytes) return launch_cfg<128, 128, 2, 2, 5, A_KC, B_KC>(p, true, true, splits, st);
                 return launch_cfg<128, 128, 2, 2, 3, A_KC, B_KC>(p, va, vb, splits, st);
        case 8: return launch_cfg<128, 128, 2, 2, 3, A_KC, B_KC>(p, va, vb, splits, st);   // = the default's fallback for unaligned operands
        case 4: if (va && vb && p.a_bytes && p.b_bytes) return launch_cfg<128, 128, 2, 2, 5, A_KC, B_KC>(p, true, true, splits, st);
                return launch_cfg<128, 128, 2, 2, 3, A_KC, B_KC>(p, va, vb, splits, st);   // buffer path: aligned, < 4 GiB operands
#ifdef LSTC_TUNING      // `make tuning` only (VERDICT r5 upkeep): the tile variants the product never selects - ~110 kernel instantiations,
                        // half of the production library's size and build time - and the timing-only ablations (wrong products by construction)
        case 1: return launch_cfg<128, 128, 2, 2, 0, A_KC, B_KC>(p, va, vb, splits, st);
        case 3: return launch_cfg<128, 128, 2, 2, 2, A_KC, B_KC>(p, va, vb, splits, st);
        case 6: return launch_cfg<256, 128, 4, 2, 2, A_KC, B_KC>(p, va, vb, splits, st);
        case 2: return launch_cfg<256, 128, 4, 2, 1, A_KC, B_KC>(p, va, vb, splits, st);
        case 5: return launch_cfg<256, 128, 4, 2, 0, A_KC, B_KC>(p, va, vb, splits, st);
        case 7: return launch_cfg<128, 128, 2, 2, 1, A_KC, B_KC>(p, va, vb, splits, st);
        case 9: return launch_cfg<256, 128, 4, 2, 3, A_KC, B_KC>(p, va, vb, splits, st);   // 8 waves x (64x64), PIPE 3
        case 10: if (va && vb && p.K % BK == 0) return launch_cfg<128, 128, 2, 2, 4, A_KC, B_KC>(p, true, true, splits, st);
                 return launch_cfg<128, 128, 2, 2, 3, A_KC, B_KC>(p, va, vb, splits, st);   // LDS-DMA needs aligned rows, full K tiles
        case 13: if constexpr (A_KC && B_KC) return launch_cfg<128, 128, 2, 2, 3 + 16 * 1, A_KC, B_KC>(p, va, vb, splits, st); return LSTC_E_UNSUPPORTED;
        case 14: if constexpr (A_KC && B_KC) return launch_cfg<128, 128, 2, 2, 3 + 16 * 3, A_KC, B_KC>(p, va, vb, splits, st); return LSTC_E_UNSUPPORTED;
        case 15: if constexpr (A_KC && B_KC) return launch_cfg<128, 128, 2, 2, 3 + 16 * 7, A_KC, B_KC>(p, va, vb, splits, st); return LSTC_E_UNSUPPORTED;
#endif
        default: return LSTC_E_UNSUPPORTED;
    }
}

}  // namespace

__attribute__((visibility("hidden"))) int lstc_gemm_f32_impl(const LstcGemmDesc* d, hipStream_t st) {
    if (!d->A || !d->B || !d->C) return LSTC_E_NULL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return LSTC_E_SHAPE;
    const int a_min = d->transA ? d->M : d->K, b_min = d->transB ? d->K : d->N;
    if (d->lda < a_min || d->ldb < b_min || d->ldc < d->N) return LSTC_E_SHAPE;
    if ((d->flags & LSTC_EPI_BIAS) && !d->bias) return LSTC_E_NULL;
    if ((d->flags & LSTC_EPI_RESIDUAL) && (!d->residual || d->ldr < d->N)) return LSTC_E_NULL;
    if ((d->flags & LSTC_EPI_RELU_MASK) && (!d->relu_src || d->ld_relu < d->N)) return LSTC_E_NULL;
    if ((d->flags & LSTC_EPI_DROPOUT) && (uint64_t)d->M * (uint64_t)d->N > 0xffffffffull) return LSTC_E_RANGE;
    if (d->transA && d->transB) return LSTC_E_UNSUPPORTED;
    const int splits = d->split_k > 1 ? d->split_k : 1;
    if (splits > 1 && d->flags != 0) return LSTC_E_UNSUPPORTED;
    GemmParams p;
    p.A = (const float*)d->A; p.B = (const float*)d->B; p.C = (float*)d->C;
    p.bias = d->bias; p.res = (const float*)d->residual; p.relu_src = (const float*)d->relu_src;
    p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.ldr = d->ldr; p.ld_relu = d->ld_relu; p.flags = d->flags; p.alpha = d->alpha;
    p.dk = make_drop_key(d->dropout_p, d->dropout_seed);
    p.batch = d->batch > 1 ? d->batch : 1;
    p.batch_stride_a = d->batch_stride_a; p.batch_stride_b = d->batch_stride_b; p.batch_stride_c = d->batch_stride_c;
    if (p.batch > 1 && (d->flags & (LSTC_EPI_BIAS | LSTC_EPI_RESIDUAL | LSTC_EPI_RELU_MASK | LSTC_EPI_DROPOUT))) return LSTC_E_UNSUPPORTED;
    if (p.batch > 65535) return LSTC_E_RANGE;
    p.ktiles = (d->K + BK - 1) / BK;
    p.ktiles_per_split = (p.ktiles + splits - 1) / splits;
#ifdef LSTC_TUNING
    p.debug = d->variant >> 4;          // tools/gemm_check ablations (no loads / no LDS writes / no barrier): wrong products
#else
    // the production library accepts the documented tile variants only: garbage in this public field must not select a
    // timing ablation or an undefined tile (LstcGemmDesc.variant, include/lstc_hip.h)
    if (!(d->variant == 0 || d->variant == 4 || d->variant == 8 || d->variant == 11 || d->variant == 12)) return LSTC_E_UNSUPPORTED;
    p.debug = 0;
#endif
    const size_t a_ext = ((size_t)((d->transA ? d->K : d->M) - 1) * d->lda + (d->transA ? d->M : d->K)) * sizeof(float) + p.batch_stride_a * sizeof(float) * (size_t)(p.batch - 1) * 0;
    const size_t b_ext = ((size_t)((d->transB ? d->N : d->K) - 1) * d->ldb + (d->transB ? d->K : d->N)) * sizeof(float);
    const bool fits32 = a_ext < 0xffffffffull && b_ext < 0xffffffffull;
    p.a_bytes = (unsigned int)(fits32 ? a_ext : 0);
    p.b_bytes = (unsigned int)(fits32 ? b_ext : 0);
    const int eff_splits = (p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;
    // float4 global loads need 16-B aligned rows; the contiguous extent must be a multiple of 4 so a
    // float4 is entirely inside or outside the matrix.
    const bool va = aligned16(d->A) && (d->lda % 4 == 0) && ((d->transA ? d->M : d->K) % 4 == 0) &&
                    (p.batch <= 1 || d->batch_stride_a % 4 == 0);
    const bool vb = aligned16(d->B) && (d->ldb % 4 == 0) && ((d->transB ? d->K : d->N) % 4 == 0) &&
                    (p.batch <= 1 || d->batch_stride_b % 4 == 0);
    p.row_off = 0;
    // float4 epilogue (epilogue_f4): one row-contiguous float4 per lane and store, operands as float4 loads - needs 16-B aligned rows
    // everywhere and at most ONE per-element operand (residual | ReLU-mask source | accumulate target)
    {
        const int naux = ((d->flags & LSTC_EPI_RESIDUAL) ? 1 : 0) + ((d->flags & LSTC_EPI_RELU_MASK) ? 1 : 0) + ((d->flags & LSTC_EPI_ACCUM) ? 1 : 0);
        const bool al = d->N % 4 == 0 && d->N >= 4 && d->ldc % 4 == 0 && aligned16(d->C) && (!(d->flags & LSTC_EPI_BIAS) || aligned16(d->bias)) &&
                        (!(d->flags & LSTC_EPI_RESIDUAL) || (d->ldr % 4 == 0 && aligned16(d->residual))) &&
                        (!(d->flags & LSTC_EPI_RELU_MASK) || (d->ld_relu % 4 == 0 && aligned16(d->relu_src)));
        // (batched launches - the split-K partials of the weight gradients - qualify when every problem's C stays 16-B aligned)
        p.epi_f4 = (al && naux <= 1 && (p.batch <= 1 || d->batch_stride_c % 4 == 0) && eff_splits == 1) ? (naux ? 2 : 1) : 0;
    }
    auto launch = [&](GemmParams& q, int variant) {
        if (!d->transA && d->transB) return launch_layout<true, true>(q, va, vb, eff_splits, variant, st);
        if (!d->transA && !d->transB) return launch_layout<true, false>(q, va, vb, eff_splits, variant, st);
        return launch_layout<false, false>(q, va, vb, eff_splits, variant, st);
    };
    // Tile-round quantisation: the default kernel runs two 128x128 workgroups per CU; a product whose tile count ends with a
    // mostly empty round of those slots (12 544 rows x 2048 columns = 1568 tiles = 3.06 rounds of 512 - one rank of the 8-GPU
    // job) pays a whole extra round.  The rows of that last round go to the 64x64-tile variant instead (4x the workgroups,
    // 4 resident per CU): every output element keeps the same k order, so the result is bit-identical to the one-launch
    // product (tests/test_hip_parity.py::test_row_split_f32_product_is_bitwise_the_single_launch_product).
    int v_main = d->variant;
    if (d->variant == 0 && eff_splits == 1 && p.batch <= 1 && !d->transA) {
        static int slots = 0;
        if (slots == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                        ? 2 * prop.multiProcessorCount : 512;
        }
        const int tM = (d->M + 127) / 128, tN = (d->N + 127) / 128;
        const long long tiles = (long long)tM * tN;
        // (The persistent walk of the same loop, variant 12, is NOT the default: with the float4 epilogue on both, one tile per workgroup
        // is faster - tools/gemm_check, 100352 x 2048 x 2048 NT: 149.6 vs 146.3 TFLOP/s plain, 146.1 vs 141.7 with dropout + residual.)
        const long long full = tiles / slots, rem = tiles % slots;
        const int main_rows = (int)((full * slots) / tN);              // whole tile rows inside the full rounds
        // (measured: a tail round up to ~30 % full gains - 4 pairs per rank 41.2 -> 40.1 ms per step; a half-full one does not)
        if (full >= 1 && rem > 0 && rem * 10 <= (long long)slots * 3 && main_rows >= 1 && main_rows < tM) {
            GemmParams pm = p, pt = p;
            const int M_main = main_rows * 128;
            pm.M = M_main;
            pt.M = d->M - M_main;
            pt.row_off = M_main;
            pt.A += (size_t)M_main * d->lda;
            pt.C += (size_t)M_main * d->ldc;
            if (pt.res) pt.res += (size_t)M_main * d->ldr;
            if (pt.relu_src) pt.relu_src += (size_t)M_main * d->ld_relu;
            int rc = launch(pm, v_main);
            if (rc) return rc;
            return launch(pt, 11);
        }
    }
    return launch(p, v_main);
}
