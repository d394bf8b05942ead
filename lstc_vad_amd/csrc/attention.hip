// Fused multi-head attention core with relative-position bias for short sequences (S <= 128), gfx950.
//
// One workgroup (4 waves) owns one (sequence, head): the whole S x S logit tile lives in LDS
// (S in {17,19,33,49,81} on this path, padded to 32*T), so there is no online softmax and no K/V
// tiling.  Both contractions run on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32):
//   scores = (Q*scale) K^T : operands straight from global in the MFMA lane layout (lane = row,
//            lane-half = a 16-float run of the 32-deep K chunk -> each lane pair reads one 128-B line);
//   O      = P V           : A operand = P from LDS (row stride 32T+1: conflict-free for lane = row and
//            for lane = column), B operand = V rows from global, 128 B per half-wave.
// Softmax / bias gather / dropout run on the LDS tile with one wave per row and wave-shuffle
// reductions.  Q, K, V, O stay in the token-major [N, S, H*d] layout the projection GEMMs produce.
// Backward recomputes nothing but the dropout mask: P is saved by the forward ([N,H,S,S], 0.6 % of
// the layer's activations), dP = dO V^T, dA = P (dP - rowsum(dP P)), dV = Pd^T dO, dQ = dA K scale,
// dK = dA^T Q scale, and the bias-table gradient is accumulated per workgroup in LDS over a chunk of
// sequences before one atomic flush.
#include "attention_common.h"

// Grid of the first / second generation kernels: one workgroup per (chunk of sequences, head).  ATTN_HEAD_FAST: the HEAD is the
// fast grid index, so the workgroups dispatched together read the 8 heads' 1-KB column slices of the SAME token rows - whole 8-KB
// rows of Q / K / V / dO per DRAM page - instead of one 1-KB slice out of every 8 KB of far-apart rows (A/B builds: 0 = chunk fast,
// the order of rounds 1-3).  gridDim.y <= 65535 bounds the chunk count in that form.
#ifndef ATTN_HEAD_FAST
#define ATTN_HEAD_FAST 0
#endif
#if ATTN_HEAD_FAST
#define ATTN_CHUNK ((int)blockIdx.y)
#define ATTN_HEAD ((int)blockIdx.x)
#define ATTN_GRID(chunks, H) dim3((unsigned)(H), (unsigned)(chunks))
#else
#define ATTN_CHUNK ((int)blockIdx.x)
#define ATTN_HEAD ((int)blockIdx.y)
#define ATTN_GRID(chunks, H) dim3((unsigned)(chunks), (unsigned)(H))
#endif

namespace {
using namespace lstc_attn;

constexpr int NT = 256;


__device__ __forceinline__ void load16(const float* __restrict__ p, int k0, int kdim, bool vec, float (&f)[16]) {
    if (vec && k0 + 16 <= kdim) {
        const float4* q = reinterpret_cast<const float4*>(p + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = q[i];
            f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] = (k0 + e < kdim) ? p[k0 + e] : 0.f;
    }
}


// 32x32 accumulator += A[32 x 16] B[32 x 16]^T where lane half h2 supplies k = 8 h2 + s (s = 0..7) of both operands.
// BF = false: eight exact-f32 v_mfma_f32_32x32x2_f32 (step s contracts k = s and k = 8 + s).  BF = true (bf16 mode,
// LstcAttnDesc.dtype = LSTC_BF16): the eight values are rounded to bf16 (RNE) and contracted by ONE v_mfma_f32_32x32x16_bf16 -
// the same k assignment, f32 accumulation; 1/16 of the matrix-core time, which leaves these kernels bound by their loads.
template <bool BF>
__device__ __forceinline__ floatx16 mma8(const float* a, const float* b, floatx16 acc) {
    if constexpr (BF) {
        attn_h8 ah, bh;
#pragma unroll
        for (int s = 0; s < 8; ++s) { ah[s] = (__bf16)a[s]; bh[s] = (__bf16)b[s]; }
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        return acc;
    }
}

// 32x32 tile of  A[rowA0.., :] * B[rowB0.., :]^T  contracted over kdim; rows beyond S-1 are clamped
// (their results are discarded by the caller).  A is scaled by a_scale before the product.
template <bool BF>
__device__ __forceinline__ floatx16 tile_abt(const float* __restrict__ A, int lda, int rowA0, const float* __restrict__ B,
                                             int ldb, int rowB0, int S, int kdim, float a_scale, bool vec) {
    const int lane = threadIdx.x & 63, r = lane & 31, h2 = lane >> 5;
    const float* pa = A + (size_t)min(rowA0 + r, S - 1) * lda;
    const float* pb = B + (size_t)min(rowB0 + r, S - 1) * ldb;
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < kdim; kb += 32) {      // not unrolled: occupancy (registers) hides the load latency, not ILP
        float a[16], b[16];
        load16(pa, kb + 16 * h2, kdim, vec, a);
        load16(pb, kb + 16 * h2, kdim, vec, b);
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] *= a_scale;
        acc = mma8<BF>(a, b, acc);
        acc = mma8<BF>(a + 8, b + 8, acc);
    }
    return acc;
}

template <int LD>
__device__ __forceinline__ void store_tile_lds(float* __restrict__ sm, int ti, int tj, const floatx16& acc) {
    const int lane = threadIdx.x & 63, c = lane & 31, h2 = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) sm[(32 * ti + (r & 3) + 8 * (r >> 2) + 4 * h2) * LD + 32 * tj + c] = acc[r];
}

// The 32-row tiles of one wave's result (MFMA 32x32 layout: lane = column c31, registers = rows) written as packed bf16 (lstc_pack1 layout,
// lstc_common.h p1_offset: `pkb` 32-k tiles per 128-row block), rows prow0 + token, k tile `ptile`.  Neighbouring lanes trade one value of each
// row pair (DPP quad_perm), so a lane holds two adjacent columns of ONE row and writes them as one dword: 64 B per row and instruction,
// half the stores and address arithmetic of a 2-byte store per value.  Tokens >= S are dropped by the buffer bounds check.
template <int T>
__device__ __forceinline__ void store_rows_packed(const floatx16 (&acc)[T], float scale, __amdgpu_buffer_rsrc_t rs, uint32_t prow0,
                                                  uint32_t ptile, uint32_t pkb, int S, int c31, int h2) {
    const uint32_t odd = (uint32_t)c31 & 1u, c2 = (uint32_t)c31 & ~1u;
    const uint32_t sel = odd ? 0x03020706u : 0x05040100u;      // even lane: (own lo, other lo) = row a; odd lane: (other hi, own hi) = row a + 1
    const uint32_t tokl = 4u * (uint32_t)h2 + odd;
    const uint32_t blk = (pkb - 1u) * 4096u;                    // element (row, k): row * 32 + (row >> 7) * (pkb - 1) * 4096 + tile * 4096 + chunk
    const uint32_t base = ptile * 4096u + (c2 & 7u);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t tok = tokl + (uint32_t)(32 * t + 2 * (j & 1) + 8 * (j >> 1));
            const uint32_t rg = prow0 + tok;
            attn_f2 f;
            f[0] = acc[t][2 * j] * scale;
            f[1] = acc[t][2 * j + 1] * scale;
            const uint32_t own = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, attn_h2));
            const uint32_t oth = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            const uint32_t val = __builtin_amdgcn_perm(oth, own, sel);
            const uint32_t e = rg * 32u + (rg >> 7) * blk + base + ((((c2 >> 3) ^ (rg >> 2)) & 3u) << 3);
            __builtin_amdgcn_raw_buffer_store_b32(val, rs, tok < (uint32_t)S ? e * 2u : 0xFFFFFFFFu, 0, 0);
        }
}

// Out[S, ncols] = scale * op(Alds) * B[S, ncols]   (op = transpose when TRANS).  Alds is SP x SP with
// stride LD and MUST be zero wherever its contraction index is >= S.  Wave w owns column tiles w, w+4, ...
// `pack` != nullptr: the result goes, rounded to bf16, into a packed [rows, K] operand (lstc_pack1 layout, `pkb` 32-k tiles per
// 128-row block) at global rows prow0 + row and k tiles ptile0 + column tile, instead of into Out.
template <int T, bool TRANS, bool BF>
__device__ __forceinline__ void lds_times_rows(const float* __restrict__ Alds, const float* __restrict__ B, int ldb, int S,
                                               int ncols, float scale, float* __restrict__ Out, int ldo,
                                               __bf16* __restrict__ pack = nullptr, uint32_t prow0 = 0, uint32_t ptile0 = 0,
                                               uint32_t pkb = 0) {
    constexpr int SP = 32 * T, LD = SP + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c31 = lane & 31, h2 = lane >> 5;
    const int ctiles = (ncols + 31) >> 5;
#pragma unroll 1
    for (int ct = wave; ct < ctiles; ct += NT / 64) {
        const int c = 32 * ct + c31;
        const bool cvalid = c < ncols;
        floatx16 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll 1
        for (int jb = 0; jb < SP; jb += 16) {
            float bv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int j = min(jb + 8 * h2 + s, S - 1);
                bv[s] = cvalid ? B[(size_t)j * ldb + c] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float av[8];
                const int i = 32 * t + c31;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int k = jb + 8 * h2 + s;
                    av[s] = TRANS ? Alds[k * LD + i] : Alds[i * LD + k];
                }
                acc[t] = mma8<BF>(av, bv, acc[t]);
            }
        }
        if (pack) {          // ncols is a multiple of 32 here (launcher), so the whole wave is valid
            store_rows_packed<T>(acc, scale, __builtin_amdgcn_make_buffer_rsrc(pack, 0, (int)0x7fffffff, 0x00020000), prow0,
                                 ptile0 + (uint32_t)ct, pkb, S, c31, h2);
        } else if (cvalid) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h2;
                    if (row >= S) continue;
                    Out[(size_t)row * ldo + c] = acc[t][r] * scale;
                }
        }
    }
}

template <int T>
__global__ void __launch_bounds__(NT, T <= 2 ? 4 : 2) attn_fwd_kernel(const AttnParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    constexpr bool BF = false;       // first generation: exact-f32 products only (bf16 products made these latency-bound loops slower)
    constexpr int SP = 32 * T, LD = SP + 1, NJ = (SP + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n = ATTN_CHUNK, h = ATTN_HEAD, S = p.S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Qb = p.Q + (size_t)n * S * p.ldq + (size_t)h * p.dk;
    const float* Kb = p.K + (size_t)n * S * p.ldk + (size_t)h * p.dk;
    const float* Vb = p.V + (size_t)n * S * p.ldv + (size_t)h * p.dv;
    float* Ob = p.O + (size_t)n * S * p.ldo + (size_t)h * p.dv;

#pragma unroll 1
    for (int t = wave; t < T * T; t += NT / 64) {
        const int ti = t / T, tj = t % T;
        if (32 * ti >= S || 32 * tj >= S) continue;   // fully padded tile: rows/cols are rewritten below
        const floatx16 acc = tile_abt<BF>(Qb, p.ldq, 32 * ti, Kb, p.ldk, 32 * tj, S, p.dk, p.scale, p.vec_qk);
        store_tile_lds<LD>(sm, ti, tj, acc);
    }
    __syncthreads();

    float* pr_base = p.probs + ((size_t)n * p.H + h) * S * S;
    const uint32_t flat0 = ((uint32_t)n * p.H + h) * (uint32_t)(S * S);
#pragma unroll 1
    for (int i = wave; i < SP; i += NT / 64) {
        float* row = sm + i * LD;
        if (i >= S) {
            for (int j = lane; j < SP; j += 64) row[j] = 0.f;
            continue;
        }
        float v[NJ];
        float m = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = lane + 64 * jj;
            float x = -INFINITY;
            if (j < S) {
                x = row[j];
                if (p.index_ld > 0 && i >= 1 && j >= 1)
                    x += p.table[(size_t)p.index[(size_t)(i - 1) * p.index_ld + (j - 1)] * p.H + h];
            }
            v[jj] = x;
            m = fmaxf(m, x);
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            v[jj] = (lane + 64 * jj < S) ? expf(v[jj] - m) : 0.f;
            s += v[jj];
        }
        s = wave_sum(s);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = lane + 64 * jj;
            if (j < SP) {
                float pv = 0.f;
                if (j < S) {
                    pv = v[jj] / s;
                    pr_base[(size_t)i * S + j] = pv;
                    if (p.has_drop) pv = drop_keep(flat0 + (uint32_t)(i * S + j), dkn) ? pv * dkn.scale : 0.f;
                }
                row[j] = pv;
            }
        }
    }
    __syncthreads();
    lds_times_rows<T, false, BF>(sm, Vb, p.ldv, S, p.dv, 1.f, Ob, p.ldo, reinterpret_cast<__bf16*>(p.Op), (uint32_t)n * (uint32_t)S,
                             (uint32_t)((h * p.dv) >> 5), (uint32_t)p.kbo);
}

template <int T>
__global__ void __launch_bounds__(NT, T <= 3 ? 2 : 1) attn_bwd_kernel(const AttnParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    constexpr bool BF = false;       // first generation: exact-f32 products only (bf16 products made these latency-bound loops slower)
    constexpr int SP = 32 * T, LD = SP + 1, NJ = (SP + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Dm = sm;                 // dP~ then dA
    float* Pm = sm + SP * LD;       // dropped probabilities
    float* tacc = sm + 2 * SP * LD; // [NT/64][table_rows] bias-table gradient of this head, one copy per wave
    const int h = ATTN_HEAD, S = p.S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool has_bias = p.index_ld > 0 && p.dtable != nullptr;
    if (has_bias)
        for (int i = threadIdx.x; i < (NT / 64) * p.table_rows; i += NT) tacc[i] = 0.f;
    // a wave owns its copy: within one row i the S - 1 columns map to distinct table rows (relative offsets of distinct
    // positions differ), and a wave walks its rows in order, so plain read-modify-write is race-free and the sum order fixed
    float* const tw = tacc + wave * p.table_rows;
    const int n_begin = ATTN_CHUNK * p.n_per_wg;
    const int n_end = min(p.N, n_begin + p.n_per_wg);
#pragma unroll 1
    for (int n = n_begin; n < n_end; ++n) {
        const float* Qb = p.Q + (size_t)n * S * p.ldq + (size_t)h * p.dk;
        const float* Kb = p.K + (size_t)n * S * p.ldk + (size_t)h * p.dk;
        const float* Vb = p.V + (size_t)n * S * p.ldv + (size_t)h * p.dv;
        const float* dOb = p.dO + (size_t)n * S * p.ldo + (size_t)h * p.dv;
        __syncthreads();   // previous sequence's LDS readers are done
#pragma unroll 1
        for (int t = wave; t < T * T; t += NT / 64) {
            const int ti = t / T, tj = t % T;
            if (32 * ti >= S || 32 * tj >= S) continue;
            const floatx16 acc = tile_abt<BF>(dOb, p.ldo, 32 * ti, Vb, p.ldv, 32 * tj, S, p.dv, 1.f, p.vec_v);
            store_tile_lds<LD>(Dm, ti, tj, acc);
        }
        __syncthreads();
        const float* pr_base = p.probs + ((size_t)n * p.H + h) * S * S;
        const uint32_t flat0 = ((uint32_t)n * p.H + h) * (uint32_t)(S * S);
#pragma unroll 1
        for (int i = wave; i < SP; i += NT / 64) {
            float* drow = Dm + i * LD;
            float* prow = Pm + i * LD;
            if (i >= S) {
                for (int j = lane; j < SP; j += 64) drow[j] = prow[j] = 0.f;
                continue;
            }
            float pv[NJ], dp[NJ], keep[NJ];
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = lane + 64 * jj;
                pv[jj] = dp[jj] = keep[jj] = 0.f;
                if (j < S) {
                    pv[jj] = pr_base[(size_t)i * S + j];
                    keep[jj] = p.has_drop ? (drop_keep(flat0 + (uint32_t)(i * S + j), dkn) ? dkn.scale : 0.f) : 1.f;
                    dp[jj] = drow[j] * keep[jj];
                    s += dp[jj] * pv[jj];
                }
            }
            s = wave_sum(s);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = lane + 64 * jj;
                if (j < SP) {
                    const float da = pv[jj] * (dp[jj] - s);      // zero for j >= S
                    drow[j] = da;
                    prow[j] = pv[jj] * keep[jj];
                    if (has_bias && i >= 1 && j >= 1 && j < S)
                        tw[p.index[(size_t)(i - 1) * p.index_ld + (j - 1)]] += da;
                }
            }
        }
        __syncthreads();
        lds_times_rows<T, true, BF>(Pm, dOb, p.ldo, S, p.dv, 1.f, p.dV + (size_t)n * S * p.ldv + (size_t)h * p.dv, p.ldv);
        lds_times_rows<T, false, BF>(Dm, Kb, p.ldk, S, p.dk, p.scale, p.dQ + (size_t)n * S * p.ldq + (size_t)h * p.dk, p.ldq);
        lds_times_rows<T, true, BF>(Dm, Qb, p.ldq, S, p.dk, p.scale, p.dK + (size_t)n * S * p.ldk + (size_t)h * p.dk, p.ldk);
    }
    if (has_bias) {
        __syncthreads();
        for (int i = threadIdx.x; i < p.table_rows; i += NT) {
            float v = tacc[i];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) v += tacc[w * p.table_rows + i];
            if (p.table_partials) p.dtable[((size_t)ATTN_CHUNK * p.table_rows + i) * p.H + h] = v;
            else atomicAdd(&p.dtable[(size_t)i * p.H + h], v);
        }
    }
}

// =====================================================================================================
// Second-generation backward (T <= 3, NW = 4 waves for T <= 2 and 8 for T = 3; d_k and d_v multiples of 32, 16-B aligned operands): same math, same LDS score tiles,
// same bias-table accumulation as attn_bwd_kernel, but the operands reach the MFMAs differently.
//   * dP = dO V^T contracts over FEATURES, so the MFMA lane layout (lane = row, 16 consecutive floats per lane) makes a
//     direct global load touch 32 different 128-B lines per instruction.  Here dO and V are staged through LDS in 32-feature
//     chunks by LDS-DMA (global_load_lds_dwordx4: 8 rows x 128 B per wave-instruction, whole lines, no staging registers),
//     double buffered, chunk c+1 in flight under the 16 MFMAs of chunk c.  Rows are 128 B in LDS; the 16-B chunk index is
//     XOR-ed with (row >> 1) & 7 on the SOURCE side of the DMA and on the read side, which puts the 16 lanes of every
//     ds_read_b128 lane group on 16 distinct 16-B slots.
//   * dV = Pd^T dO, dQ = dA K, dK = dA^T Q contract over the S tokens: lane = output column, so global loads are coalesced
//     already; each wave walks a list of (product, 32-column tile) jobs, holds ALL S rows of the job's B operand in
//     registers (SP/2 floats per lane) and requests the next job's rows before the 32 T MFMAs of the current one.
// First version (attn_bwd_kernel): every K chunk exposed one global-load latency; 22 % MFMA-busy, 2.8 ms per LTN layer.
// Phase 3 of the backward as ONE software pipeline over (product, 32-column tile) jobs.  Every product is
// Out[S, cols] = scale * A^T B with A^T read from an LDS tile laid out [k][i] (the caller keeps dA both ways), B = rows of a
// global matrix.  Per job a lane keeps the SP / 2 values of its column that its MFMA lane-half consumes (rows 16 b + 8 h2 + s)
// in registers; the rows of the wave's NEXT job - possibly of the next product - are requested before the 32 T MFMAs of the
// current one (two register sets, loop unrolled by two so no set is ever copied).
//   * buffer loads / stores: ONE per-lane offset register (column + the lane-half's row shift), the row rides in the scalar
//     offset, and rows >= S fall outside the descriptors' S * ld * 4 bytes: they read 0 and their stores are dropped - no
//     clamps, no 64-bit per-access addresses (plain pointers: 2 address registers per value, 682 spilled registers);
//   * macros, not lambdas taking the register arrays by reference (those sent both sets to scratch).
struct RtlJob {
    const float* A;                    // LDS tile, [k][i] with row stride LD
    __amdgpu_buffer_rsrc_t b, o;       // B rows, output rows (S rows each)
    uint32_t brow, orow;               // row pitches in bytes
    float scale;
    int ct;                            // 32-column tile
    int pk;                            // output as packed bf16: o = the whole pack, ptile = its k tile, pkb = tiles per row block
    uint32_t ptile, pkb;
};
#define RTL_LOAD(J, bv_)                                                                                      \
    do {                                                                                                      \
        const uint32_t vo_ = (uint32_t)c31 * 4u + (uint32_t)(8 * h2) * J.brow + (uint32_t)(128 * J.ct);       \
        _Pragma("unroll") for (int b_ = 0; b_ < SP / 16; ++b_)                                                  \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_)                                                    \
                bv_[8 * b_ + s_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(             \
                    J.b, vo_, (uint32_t)(16 * b_ + s_) * J.brow, 0));                                         \
    } while (0)
#define RTL_COMPUTE(J, bv_)                                                                                   \
    do {                                                                                                      \
        floatx16 o_[T];                                                                                       \
        _Pragma("unroll") for (int t_ = 0; t_ < T; ++t_)                                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) o_[t_][i_] = 0.f;                                 \
        _Pragma("unroll") for (int b_ = 0; b_ < SP / 16; ++b_) {                                                \
            _Pragma("unroll") for (int t_ = 0; t_ < T; ++t_) {                                                  \
                float av_[8];                                                                                 \
                _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_)                                                \
                    av_[s_] = J.A[(16 * b_ + 8 * h2 + s_) * LD + 32 * t_ + c31];                              \
                o_[t_] = mma8<BF>(av_, &bv_[8 * b_], o_[t_]);                                                  \
            }                                                                                                 \
        }                                                                                                     \
        if (J.pk) {     /* packed bf16 rows (store_rows_packed) */                                              \
            store_rows_packed<T>(o_, J.scale, J.o, prow0, J.ptile, J.pkb, S, c31, h2);                          \
        } else {                                                                                              \
            const uint32_t wo_ = (uint32_t)c31 * 4u + (uint32_t)(4 * h2) * J.orow + (uint32_t)(128 * J.ct);   \
            _Pragma("unroll") for (int t_ = 0; t_ < T; ++t_)                                                    \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_)                                               \
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o_[t_][r_] * J.scale), J.o, wo_, \
                        (uint32_t)(32 * t_ + (r_ & 3) + 8 * (r_ >> 2)) * J.orow, 0);                          \
        }                                                                                                     \
    } while (0)

#undef RTL_PLACEHOLDER
struct StageDma {
    // one DMA piece = 8 rows x 32 floats (1 KB); lane -> (row in piece, 16-B position); the source chunk is swizzled
    __device__ static __forceinline__ void issue(const float* __restrict__ base, int ld, int S, int row0, int kc, uint32_t lds_bytes) {
        const int lane = threadIdx.x & 63;
        const int r = min(row0 + (lane >> 3), S - 1), c = lane & 7;
        const float* g = base + (size_t)r * ld + kc * 32 + ((c ^ ((r >> 1) & 7)) << 2);
        const uint32_t lb = __builtin_amdgcn_readfirstlane(lds_bytes);        // wave-uniform by construction; make it provable
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lb) : "memory");
    }
};

template <int T, bool BF, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) attn_bwd2_kernel(const AttnParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    constexpr int SP = 32 * T, LD = SP + 1, NJ = (SP + 63) / 64;
    constexpr int CH = SP * 32;                  // floats of one staged operand chunk (SP rows x 32 features)
    constexpr int NPW = 2 * (SP / 8) / NW;                       // DMA pieces per wave and chunk: 2 operands x SP/8 pieces / 4 waves
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Dm = sm;                              // dP~ then dA
    float* Pm = sm + SP * LD;                    // dropped probabilities
    constexpr int ST0 = (2 * SP * LD + 3) & ~3;  // staging ring: [buf][operand][SP][32], 16-B aligned
    float* stage = sm + ST0;
    float* DmT = stage;                          // dA^T (phase 2 / 3): the staging ring is idle after phase 1 (SP * LD <= 4 * CH)
    float* tacc = stage + 4 * CH;                // [NW][table_rows]
    const int h = ATTN_HEAD, S = p.S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const bool has_bias = p.index_ld > 0 && p.dtable != nullptr;
    if (has_bias)
        for (int i = threadIdx.x; i < NW * p.table_rows; i += 64 * NW) tacc[i] = 0.f;
    float* const tw = tacc + wave * p.table_rows;
    const int n_begin = ATTN_CHUNK * p.n_per_wg;
    const int n_end = min(p.N, n_begin + p.n_per_wg);
    const uint32_t stage_b = (uint32_t)(ST0 * 4);       // dynamic LDS starts at byte 0 of the workgroup's allocation
    const int nchunks = p.dv >> 5;
#pragma unroll 1
    for (int n = n_begin; n < n_end; ++n) {
        const float* Qb = p.Q + (size_t)n * S * p.ldq + (size_t)h * p.dk;
        const float* Kb = p.K + (size_t)n * S * p.ldk + (size_t)h * p.dk;
        const float* Vb = p.V + (size_t)n * S * p.ldv + (size_t)h * p.dv;
        const float* dOb = p.dO + (size_t)n * S * p.ldo + (size_t)h * p.dv;
        __syncthreads();   // previous sequence's LDS readers are done (plain barrier: no DMA is in flight here)
        // ---- phase 1: dP~ = dO V^T, operands staged per 32-feature chunk
        auto issue_chunk = [&](int kc, int buf) {
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int piece = wave + NW * j;                  // 0 .. 2*SP/8-1: first half dO, second half V
                const int op = piece >= SP / 8 ? 1 : 0, pj = piece - op * (SP / 8);
                StageDma::issue(op ? Vb : dOb, op ? p.ldv : p.ldo, S, 8 * pj, kc, stage_b + (uint32_t)(((buf * 2 + op) * CH + pj * 256) * 4));
            }
        };
        constexpr int NA = (T * T + NW - 1) / NW;     // 32x32 score tiles per wave
        floatx16 acc[NA];
#pragma unroll
        for (int t = 0; t < NA; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        issue_chunk(0, 0);
        // the probabilities of this wave's rows (phase 2) are requested now and land under phase 1 (first version: one exposed
        // global-load latency per row, 16 rows per wave)
        const float* pr_base = p.probs + ((size_t)n * p.H + h) * S * S;
        float pvr[SP / NW][NJ];
        int idxr[SP / NW][NJ];                 // bias-table row of (i, j), -1 where no bias applies
#pragma unroll
        for (int ii = 0; ii < SP / NW; ++ii)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int i = wave + NW * ii, j = lane + 64 * jj;
                pvr[ii][jj] = (i < S && j < S) ? pr_base[(size_t)i * S + j] : 0.f;
                idxr[ii][jj] = (has_bias && i >= 1 && j >= 1 && i < S && j < S) ? (int)p.index[(size_t)(i - 1) * p.index_ld + (j - 1)] : -1;
            }
#pragma unroll 1
        for (int kc = 0; kc < nchunks; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < nchunks) {
                issue_chunk(kc + 1, buf ^ 1);
                __builtin_amdgcn_s_waitcnt((NPW & 15) | (7 << 4) | (15 << 8) | ((NPW >> 4) << 14));    // vmcnt(NPW): chunk kc landed
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);                                                    // vmcnt(0)
            }
            __builtin_amdgcn_s_barrier();                        // every wave's pieces of chunk kc are in LDS
            const float* sA = stage + (buf * 2 + 0) * CH;
            const float* sB = stage + (buf * 2 + 1) * CH;
#pragma unroll
            for (int tt = 0; tt < NA; ++tt) {
                const int t = wave + NW * tt;
                if (t < T * T) {
                    const int ra = 32 * (t / T) + l31, rb = 32 * (t % T) + l31;
                    float a[16], b[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 va = *reinterpret_cast<const float4*>(sA + ra * 32 + (((4 * h2 + q) ^ ((ra >> 1) & 7)) << 2));
                        const float4 vb = *reinterpret_cast<const float4*>(sB + rb * 32 + (((4 * h2 + q) ^ ((rb >> 1) & 7)) << 2));
                        a[4 * q] = va.x; a[4 * q + 1] = va.y; a[4 * q + 2] = va.z; a[4 * q + 3] = va.w;
                        b[4 * q] = vb.x; b[4 * q + 1] = vb.y; b[4 * q + 2] = vb.z; b[4 * q + 3] = vb.w;
                    }
                    acc[tt] = mma8<BF>(a, b, acc[tt]);
                    acc[tt] = mma8<BF>(a + 8, b + 8, acc[tt]);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): this chunk's fragments are in registers
            __builtin_amdgcn_s_barrier();                        // buffer kc & 1 may be refilled (chunk kc + 2)
        }
#pragma unroll
        for (int tt = 0; tt < NA; ++tt) {
            const int t = wave + NW * tt;
            if (t < T * T) store_tile_lds<LD>(Dm, t / T, t % T, acc[tt]);
        }
        __syncthreads();
        // ---- phase 2: dA = P (dP - rowsum(dP P)), bias-table gradient (as attn_bwd_kernel)
        const uint32_t flat0 = ((uint32_t)n * p.H + h) * (uint32_t)(S * S);
#pragma unroll
        for (int ii = 0; ii < SP / NW; ++ii) {
            const int i = wave + NW * ii;
            float* drow = Dm + i * LD;
            float* prow = Pm + i * LD;
            if (i >= S) {
                for (int j = lane; j < SP; j += 64) { drow[j] = prow[j] = 0.f; DmT[j * LD + i] = 0.f; }
                continue;
            }
            float pv[NJ], dp[NJ], keep[NJ];
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = lane + 64 * jj;
                pv[jj] = dp[jj] = keep[jj] = 0.f;
                if (j < S) {
                    pv[jj] = pvr[ii][jj];
                    keep[jj] = p.has_drop ? (drop_keep(flat0 + (uint32_t)(i * S + j), dkn) ? dkn.scale : 0.f) : 1.f;
                    dp[jj] = drow[j] * keep[jj];
                    s += dp[jj] * pv[jj];
                }
            }
            s = wave_sum(s);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = lane + 64 * jj;
                if (j < SP) {
                    const float da = pv[jj] * (dp[jj] - s);      // zero for j >= S
                    drow[j] = da;
                    DmT[j * LD + i] = da;
                    prow[j] = pv[jj] * keep[jj];
                    if (idxr[ii][jj] >= 0) tw[idxr[ii][jj]] += da;
                }
            }
        }
        __syncthreads();
        // ---- phase 3: dV = Pd^T dO, dQ = scale (dA^T)^T K, dK = scale dA^T Q: one pipeline over this wave's (product, column
        // tile) jobs; dV's tiles first, then dQ's, then dK's
        {
            const int jv = ((p.dv >> 5) - wave + NW - 1) / NW, jk = ((p.dk >> 5) - wave + NW - 1) / NW;      // tiles of this wave
            const int njobs = jv + 2 * jk;
            const uint32_t bytes_v = (uint32_t)S * (uint32_t)p.ldv * 4u, bytes_o = (uint32_t)S * (uint32_t)p.ldo * 4u;
            const uint32_t bytes_q = (uint32_t)S * (uint32_t)p.ldq * 4u, bytes_k = (uint32_t)S * (uint32_t)p.ldk * 4u;
            const __amdgpu_buffer_rsrc_t r_dO = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dOb), 0, (int)bytes_o, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_K = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Kb), 0, (int)bytes_k, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_Q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Qb), 0, (int)bytes_q, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_dV = __builtin_amdgcn_make_buffer_rsrc(p.dV + (size_t)n * S * p.ldv + (size_t)h * p.dv, 0, (int)bytes_v, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_dQ = __builtin_amdgcn_make_buffer_rsrc(p.dQ + (size_t)n * S * p.ldq + (size_t)h * p.dk, 0, (int)bytes_q, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_dK = __builtin_amdgcn_make_buffer_rsrc(p.dK + (size_t)n * S * p.ldk + (size_t)h * p.dk, 0, (int)bytes_k, 0x00020000);
            const bool pk = p.dQp != nullptr;
            const uint32_t prow0 = (uint32_t)n * (uint32_t)S;                  // global row of the sequence's first token
            const __amdgpu_buffer_rsrc_t r_dVp = __builtin_amdgcn_make_buffer_rsrc(pk ? p.dVp : (void*)p.dV, 0, (int)0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_dQp = __builtin_amdgcn_make_buffer_rsrc(pk ? p.dQp : (void*)p.dQ, 0, (int)0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_dKp = __builtin_amdgcn_make_buffer_rsrc(pk ? p.dKp : (void*)p.dK, 0, (int)0x7fffffff, 0x00020000);
            auto job = [&](int kj) -> RtlJob {
                RtlJob J;
                J.pk = pk ? 1 : 0;
                if (kj < jv) { J.A = Pm; J.b = r_dO; J.brow = (uint32_t)p.ldo * 4u; J.o = pk ? r_dVp : r_dV; J.orow = (uint32_t)p.ldv * 4u; J.scale = 1.f; J.ct = wave + NW * kj;
                               J.pkb = (uint32_t)p.kbv; J.ptile = (uint32_t)(p.tv0 + ((h * p.dv) >> 5)) + (uint32_t)J.ct; }
                else if (kj < jv + jk) { J.A = DmT; J.b = r_K; J.brow = (uint32_t)p.ldk * 4u; J.o = pk ? r_dQp : r_dQ; J.orow = (uint32_t)p.ldq * 4u; J.scale = p.scale; J.ct = wave + NW * (kj - jv);
                               J.pkb = (uint32_t)p.kbq; J.ptile = (uint32_t)(p.tq0 + ((h * p.dk) >> 5)) + (uint32_t)J.ct; }
                else { J.A = Dm; J.b = r_Q; J.brow = (uint32_t)p.ldq * 4u; J.o = pk ? r_dKp : r_dK; J.orow = (uint32_t)p.ldk * 4u; J.scale = p.scale; J.ct = wave + NW * (kj - jv - jk);
                               J.pkb = (uint32_t)p.kbk; J.ptile = (uint32_t)(p.tk0 + ((h * p.dk) >> 5)) + (uint32_t)J.ct; }
                return J;
            };
            constexpr int RB = SP / 2;
            const int c31 = l31;
            float b0[RB], b1[RB];
            RtlJob J0 = job(0), J1 = job(1);
            if (njobs > 0) RTL_LOAD(J0, b0);
#pragma unroll 1
            for (int kj = 0; kj < njobs; kj += 2) {
                if (kj + 1 < njobs) RTL_LOAD(J1, b1);
                RTL_COMPUTE(J0, b0);
                if (kj + 1 < njobs) {
                    J0 = job(kj + 2);
                    if (kj + 2 < njobs) RTL_LOAD(J0, b0);
                    RTL_COMPUTE(J1, b1);
                    J1 = job(kj + 3);
                }
            }
        }
    }
    if (has_bias) {
        __syncthreads();
        for (int i = threadIdx.x; i < p.table_rows; i += 64 * NW) {
            float v = tacc[i];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += tacc[w * p.table_rows + i];
            if (p.table_partials) p.dtable[((size_t)ATTN_CHUNK * p.table_rows + i) * p.H + h] = v;
            else atomicAdd(&p.dtable[(size_t)i * p.H + h], v);
        }
    }
}

// Second-generation forward: Q K^T with both operands staged by LDS-DMA (as dO V^T in the backward), softmax as in
// attn_fwd_kernel but writing the dropped probabilities BOTH ways (row-major for nothing but symmetry with the backward is not
// needed: only the [k][i] copy feeds P V), then O = Pd V as the job pipeline of the backward (A = Pd^T tile, B = V rows).
template <int T, bool BF, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) attn_fwd2_kernel(const AttnParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    constexpr int SP = 32 * T, LD = SP + 1, NJ = (SP + 63) / 64;
    constexpr int CH = SP * 32;
    constexpr int NPW = 2 * (SP / 8) / NW;    
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Am = sm;                              // logits, then probabilities (row-major)
    constexpr int ST0 = (SP * LD + 3) & ~3;
    float* stage = sm + ST0;                     // staging ring [buf][operand][SP][32]; after phase 1: Pd^T ([k][i])
    float* PT = stage;
    const int n = ATTN_CHUNK, h = ATTN_HEAD, S = p.S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const float* Qb = p.Q + (size_t)n * S * p.ldq + (size_t)h * p.dk;
    const float* Kb = p.K + (size_t)n * S * p.ldk + (size_t)h * p.dk;
    const float* Vb = p.V + (size_t)n * S * p.ldv + (size_t)h * p.dv;
    const uint32_t stage_b = (uint32_t)(ST0 * 4);
    const int nchunks = p.dk >> 5;
    auto issue_chunk = [&](int kc, int buf) {
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int piece = wave + NW * j;
            const int op = piece >= SP / 8 ? 1 : 0, pj = piece - op * (SP / 8);
            StageDma::issue(op ? Kb : Qb, op ? p.ldk : p.ldq, S, 8 * pj, kc, stage_b + (uint32_t)(((buf * 2 + op) * CH + pj * 256) * 4));
        }
    };
    constexpr int NA = (T * T + NW - 1) / NW;     // 32x32 score tiles per wave
    floatx16 acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    issue_chunk(0, 0);
    // relative-position bias of this wave's softmax rows: table[index[i-1, j-1], h] is two DEPENDENT global loads per element;
    // requested here they land under the Q K^T phase (first version: in the row loop, one exposed double latency per row)
    float biasr[SP / NW][NJ];
#pragma unroll
    for (int ii = 0; ii < SP / NW; ++ii)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int i = wave + NW * ii, j = lane + 64 * jj;
            biasr[ii][jj] = (p.index_ld > 0 && i >= 1 && j >= 1 && i < S && j < S)
                                ? p.table[(size_t)p.index[(size_t)(i - 1) * p.index_ld + (j - 1)] * p.H + h] : 0.f;
        }
#pragma unroll 1
    for (int kc = 0; kc < nchunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunks) {
            issue_chunk(kc + 1, buf ^ 1);
            __builtin_amdgcn_s_waitcnt((NPW & 15) | (7 << 4) | (15 << 8) | ((NPW >> 4) << 14));
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        __builtin_amdgcn_s_barrier();
        const float* sA = stage + (buf * 2 + 0) * CH;
        const float* sB = stage + (buf * 2 + 1) * CH;
#pragma unroll
        for (int tt = 0; tt < NA; ++tt) {
            const int t = wave + NW * tt;
            if (t < T * T) {
                const int ra = 32 * (t / T) + l31, rb = 32 * (t % T) + l31;
                float a[16], b[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 va = *reinterpret_cast<const float4*>(sA + ra * 32 + (((4 * h2 + q) ^ ((ra >> 1) & 7)) << 2));
                    const float4 vb = *reinterpret_cast<const float4*>(sB + rb * 32 + (((4 * h2 + q) ^ ((rb >> 1) & 7)) << 2));
                    a[4 * q] = va.x; a[4 * q + 1] = va.y; a[4 * q + 2] = va.z; a[4 * q + 3] = va.w;
                    b[4 * q] = vb.x; b[4 * q + 1] = vb.y; b[4 * q + 2] = vb.z; b[4 * q + 3] = vb.w;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) a[s] *= p.scale;
                acc[tt] = mma8<BF>(a, b, acc[tt]);
                acc[tt] = mma8<BF>(a + 8, b + 8, acc[tt]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int tt = 0; tt < NA; ++tt) {
        const int t = wave + NW * tt;
        if (t < T * T) store_tile_lds<LD>(Am, t / T, t % T, acc[tt]);
    }
    __syncthreads();
    // ---- softmax rows (as attn_fwd_kernel); the dropped probabilities go to PT[j][i] for the P V product
    float* pr_base = p.probs + ((size_t)n * p.H + h) * S * S;
    const uint32_t flat0 = ((uint32_t)n * p.H + h) * (uint32_t)(S * S);
#pragma unroll
    for (int ii = 0; ii < SP / NW; ++ii) {
        const int i = wave + NW * ii;
        float* row = Am + i * LD;
        if (i >= S) {
            for (int j = lane; j < SP; j += 64) PT[j * LD + i] = 0.f;
            continue;
        }
        float v[NJ];
        float m = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = lane + 64 * jj;
            float x = -INFINITY;
            if (j < S) x = row[j] + biasr[ii][jj];
            v[jj] = x;
            m = fmaxf(m, x);
        }
        m = wave_max(m);
        float s = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            v[jj] = (lane + 64 * jj < S) ? expf(v[jj] - m) : 0.f;
            s += v[jj];
        }
        s = wave_sum(s);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = lane + 64 * jj;
            if (j < SP) {
                float pv = 0.f;
                if (j < S) {
                    pv = v[jj] / s;
                    pr_base[(size_t)i * S + j] = pv;
                    if (p.has_drop) pv = drop_keep(flat0 + (uint32_t)(i * S + j), dkn) ? pv * dkn.scale : 0.f;
                }
                PT[j * LD + i] = pv;
            }
        }
    }
    __syncthreads();
    // ---- O = Pd V: job pipeline over this wave's 32-column tiles
    {
        const int njobs = ((p.dv >> 5) - wave + NW - 1) / NW;
        const uint32_t bytes_v = (uint32_t)S * (uint32_t)p.ldv * 4u, bytes_o = (uint32_t)S * (uint32_t)p.ldo * 4u;
        const __amdgpu_buffer_rsrc_t r_V = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vb), 0, (int)bytes_v, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_O = __builtin_amdgcn_make_buffer_rsrc(p.O + (size_t)n * S * p.ldo + (size_t)h * p.dv, 0, (int)bytes_o, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_Op = __builtin_amdgcn_make_buffer_rsrc(p.Op ? p.Op : (void*)p.O, 0, (int)0x7fffffff, 0x00020000);
        auto job = [&](int kj) -> RtlJob {
            RtlJob J;
            J.A = PT; J.b = r_V; J.brow = (uint32_t)p.ldv * 4u; J.o = p.Op ? r_Op : r_O; J.orow = (uint32_t)p.ldo * 4u; J.scale = 1.f; J.ct = wave + NW * kj;
            J.pk = p.Op ? 1 : 0; J.ptile = (uint32_t)((h * p.dv) >> 5) + (uint32_t)J.ct; J.pkb = (uint32_t)p.kbo;
            return J;
        };
        const uint32_t prow0 = (uint32_t)n * (uint32_t)S;
        constexpr int RB = SP / 2;
        const int c31 = l31;
        float b0[RB], b1[RB];
        RtlJob J0 = job(0), J1 = job(1);
        if (njobs > 0) RTL_LOAD(J0, b0);
#pragma unroll 1
        for (int kj = 0; kj < njobs; kj += 2) {
            if (kj + 1 < njobs) RTL_LOAD(J1, b1);
            RTL_COMPUTE(J0, b0);
            if (kj + 1 < njobs) {
                J0 = job(kj + 2);
                if (kj + 2 < njobs) RTL_LOAD(J0, b0);
                RTL_COMPUTE(J1, b1);
                J1 = job(kj + 3);
            }
        }
    }
}

#undef RTL_LOAD
#undef RTL_COMPUTE

int fill_params(const LstcAttnDesc* d, AttnParams& p, bool bwd) {
    if (!d) return LSTC_E_NULL;
    if (d->dtype != LSTC_F32 && d->dtype != LSTC_BF16) return LSTC_E_UNSUPPORTED;
    if (!d->Q || !d->K || !d->V || !d->probs) return LSTC_E_NULL;
    if (!bwd && !d->O && !d->O_pack) return LSTC_E_NULL;
    const bool gpk = bwd && d->dQ_pack && d->dK_pack && d->dV_pack;
    if (bwd && (!d->dO || (!gpk && (!d->dQ || !d->dK || !d->dV)))) return LSTC_E_NULL;
    if (d->N <= 0 || d->S < 1 || d->H <= 0 || d->dk <= 0 || d->dv <= 0) return LSTC_E_SHAPE;
    if (d->S > 128) return LSTC_E_RANGE;
    if (d->in_pack_cols < 0 || d->dO_pack_cols < 0) return LSTC_E_SHAPE;
    if (d->in_pack_cols == 0 && (d->ldq < d->H * d->dk || d->ldk < d->H * d->dk || d->ldv < d->H * d->dv)) return LSTC_E_SHAPE;
    if (!(d->O_pack && !bwd) && !(bwd && d->dO_pack_cols > 0) && d->ldo < d->H * d->dv) return LSTC_E_SHAPE;
    if ((uint64_t)d->N * d->H * d->S * d->S > 0xffffffffull) return LSTC_E_RANGE;
    if (d->index_ld > 0 && (!d->table || !d->index || d->index_ld < d->S - 1)) return LSTC_E_NULL;
    p.Q = (const float*)d->Q; p.K = (const float*)d->K; p.V = (const float*)d->V; p.O = (float*)d->O;
    p.probs = d->probs; p.table = d->table; p.index = d->index;
    p.dO = (const float*)d->dO; p.dQ = (float*)d->dQ; p.dK = (float*)d->dK; p.dV = (float*)d->dV; p.dtable = d->dtable;
    p.N = d->N; p.S = d->S; p.H = d->H; p.dk = d->dk; p.dv = d->dv;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo; p.index_ld = d->index_ld;
    p.table_rows = 0;
    p.scale = d->scale;
    p.has_drop = d->dropout_p > 0.f;
    p.dkey = make_drop_key(d->dropout_p, d->dropout_seed);
    p.vec_qk = d->dk % 4 == 0 && d->ldq % 4 == 0 && d->ldk % 4 == 0 && aligned16(d->Q) && aligned16(d->K);
    p.vec_v = d->dv % 4 == 0 && d->ldv % 4 == 0 && d->ldo % 4 == 0 && aligned16(d->V) && (!bwd || aligned16(d->dO));
    p.n_per_wg = 1;
    p.dQp = p.dKp = p.dVp = nullptr;
    p.kbq = p.kbk = p.kbv = 0;
    p.tq0 = p.tk0 = p.tv0 = 0;
    p.Op = nullptr; p.kbo = 0;
    p.Qi = p.Ki = p.Vi = p.dOi = nullptr;
    p.kiq = p.kik = p.kiv = p.kido = 0;
    p.iq0 = p.ik0 = p.iv0 = p.ido0 = 0;
    p.pld = d->S;
    if (d->probs_ld != 0 && d->probs_ld != d->S && d->in_pack_cols <= 0) return LSTC_E_UNSUPPORTED;
    return 0;
}

// Packed-input form (LstcAttnDesc.in_pack_cols > 0): shape / alignment checks shared by forward and backward, pack geometry.
int fill_packed_inputs(const LstcAttnDesc* d, AttnParams& p, bool bwd) {
    const int64_t M = (int64_t)p.N * p.S;
    if (d->dtype != LSTC_BF16 || p.S > 96 || p.dk % 32 || p.dv % 64 || M % 256) return LSTC_E_UNSUPPORTED;
    if (d->in_pack_cols % 64 || d->Q_col0 % 32 || d->K_col0 % 32 || d->V_col0 % 32 || d->Q_col0 < 0 || d->K_col0 < 0 || d->V_col0 < 0 ||
        d->Q_col0 + p.H * p.dk > d->in_pack_cols || d->K_col0 + p.H * p.dk > d->in_pack_cols || d->V_col0 + p.H * p.dv > d->in_pack_cols ||
        M * (int64_t)d->in_pack_cols * 2 > 0x7fffffffLL) return LSTC_E_SHAPE;
    if (!aligned16(d->Q) || !aligned16(d->K) || !aligned16(d->V) || !aligned16(d->probs)) return LSTC_E_ALIGN;
    if (d->probs_ld < p.S || d->probs_ld % 4 || (int64_t)p.S * d->probs_ld * 4 > 0x7fffffffLL) return LSTC_E_SHAPE;
    p.pld = d->probs_ld;
    p.Qi = (const __bf16*)d->Q; p.Ki = (const __bf16*)d->K; p.Vi = (const __bf16*)d->V;
    p.kiq = p.kik = p.kiv = d->in_pack_cols / 32;
    p.iq0 = d->Q_col0 / 32; p.ik0 = d->K_col0 / 32; p.iv0 = d->V_col0 / 32;
    if (bwd) {
        if (d->dO_pack_cols <= 0) return LSTC_E_UNSUPPORTED;
        if (d->dO_pack_cols % 64 || d->dO_col0 % 32 || d->dO_col0 < 0 || d->dO_col0 + p.H * p.dv > d->dO_pack_cols ||
            M * (int64_t)d->dO_pack_cols * 2 > 0x7fffffffLL) return LSTC_E_SHAPE;
        if (!aligned16(d->dO)) return LSTC_E_ALIGN;
        p.dOi = (const __bf16*)d->dO;
        p.kido = d->dO_pack_cols / 32;
        p.ido0 = d->dO_col0 / 32;
    }
    return 0;
}

template <typename Kern>
void set_lds(Kern k, size_t lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace

extern "C" {

int lstc_attn_fwd(const LstcAttnDesc* d, void* stream) {
    AttnParams p;
    int rc = fill_params(d, p, false);
    if (rc) return rc;
    if (d->O_pack) {        // packed bf16 output: token rows and head columns fill the pack's even tile grid exactly
        const int64_t M = (int64_t)p.N * p.S;
        if (M % 256 || (p.H * p.dv) % 64 || p.dv % 32 || M * (int64_t)(p.H * p.dv) * 2 > 0x7fffffffLL) return LSTC_E_UNSUPPORTED;
        if (!aligned16(d->O_pack)) return LSTC_E_ALIGN;
        p.Op = d->O_pack;
        p.kbo = (p.H * p.dv) / 32;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool bf = d->dtype == LSTC_BF16;
    const int T = (p.S + 31) / 32;
    if (d->in_pack_cols > 0) {      // third generation: packed bf16 inputs and output
        if (!d->O_pack) return LSTC_E_UNSUPPORTED;
        rc = fill_packed_inputs(d, p, false);
        if (rc) return rc;
        // sequences per workgroup: a workgroup's ring runs across them (no cold start per sequence); ~2 rounds of the 512
        // resident workgroups at T >= 2 (measured, 16384 items: S = 81 0.81 / 0.71 / 0.65 / 0.63 ms at 1 / 2 / 4 / 8), more and
        // smaller workgroups at T = 1 (1536 resident; S = 17 0.113 / 0.119 / 0.134 ms at 1 / 2 / 8)
        int npw = (int)(((int64_t)p.N * p.H) / (T == 1 ? 8192 : 1024));
        if (d->variant >= 100) npw = d->variant - 100;          // measurement hook
        npw = npw < 1 ? 1 : (npw > 16 ? 16 : npw);
        p.n_per_wg = npw;
        dim3 grid3((p.N + npw - 1) / npw, p.H);
        return attn3_fwd_launch(p, T, (int)grid3.x, st);
    }
    if (!bf && (d->variant == 3 || (d->variant == 0 && T != 2)) && d->O && T <= 3 && p.vec_qk && p.vec_v && p.dk % 32 == 0 && p.dv % 64 == 0) {
        // third-generation structure on the exact-f32 MFMA (csrc/attention_pk.hip, attn_fwd3f_kernel).  Same box, N = 2048, H = 8,
        // d_k = 256 (tools/attn3f_check.py): S = 81 2.85 -> 1.89 ms, S = 96 3.1 -> 2.1, S = 17 0.30 -> 0.27; S = 49 1.13 vs 1.17 and
        // S = 33 0.97 vs 1.1 (the padded 64-key tiles cost MFMA time the first generation skips): 32 < S <= 64 stays where it was
        int npw = (int)(((int64_t)p.N * p.H) / (T == 1 ? 8192 : 1024));
        npw = npw < 1 ? 1 : (npw > 16 ? 16 : npw);
        p.n_per_wg = npw;
        return attn3f_fwd_launch(p, T, (p.N + npw - 1) / npw, st);
    }
    const size_t lds = (size_t)(32 * T) * (32 * T + 1) * sizeof(float);
    dim3 grid = ATTN_GRID(p.N, p.H);
    if ((T == 1 || T == 3 || (bf && T == 2)) && p.vec_qk && p.vec_v && (p.dk % 32) == 0 && (p.dv % 32) == 0 && (d->variant == 0 || d->variant == 2)) {
        // second-generation kernel (LDS-DMA staged Q K^T, register-resident V rows).  Interleaved A/B on one MI355X
        // (tools/attn_time.py): S = 17 0.283 vs 0.350 ms; S = 49 with exact-f32 products 1.10 vs 1.02 ms (the first generation's 4
        // waves per SIMD hide more latency than this kernel's 2) - stays on the first generation; with bf16 products (LSTC_BF16)
        // the staged kernel wins there too (0.87 vs 1.16 ms).  64 < S <= 96 needs 86 KB of LDS (one workgroup per CU): that
        // instantiation runs 8 waves.
        const int SP = 32 * T;
        const size_t lds2 = ((size_t)((SP * (SP + 1) + 3) & ~3) + (size_t)4 * SP * 32) * sizeof(float);
#define LSTC_FWD2(TT, BB, WW)                                                              \
    do {                                                                                   \
        static LstcDevOnce once2;                                                          \
        const int dev2_ = once2.begin();                                                   \
        if (dev2_ >= 0) { set_lds(attn_fwd2_kernel<TT, BB, WW>, 160 * 1024); once2.end(dev2_); } \
        hipLaunchKernelGGL((attn_fwd2_kernel<TT, BB, WW>), grid, 64 * WW, lds2, st, p);    \
    } while (0)
        if (!bf) { if (T == 1) LSTC_FWD2(1, false, 4); else LSTC_FWD2(3, false, 8); }
        else if (T == 1) LSTC_FWD2(1, true, 4); else if (T == 2) LSTC_FWD2(2, true, 4); else LSTC_FWD2(3, true, 8);
#undef LSTC_FWD2
        return lstc_launch_status();
    }
#define LSTC_FWD(TT)                                                      \
    do {                                                                  \
        static LstcDevOnce once;                                          \
        const int dev_ = once.begin();                                    \
        if (dev_ >= 0) { set_lds(attn_fwd_kernel<TT>, 160 * 1024); once.end(dev_); } \
        hipLaunchKernelGGL(attn_fwd_kernel<TT>, grid, NT, lds, st, p);    \
    } while (0)
    switch (T) {
        case 1: LSTC_FWD(1); break;
        case 2: LSTC_FWD(2); break;
        case 3: LSTC_FWD(3); break;
        default: LSTC_FWD(4); break;
    }
#undef LSTC_FWD
    return lstc_launch_status();
}

// d->dtable rows: the caller passes the table row count through `index_ld`'s companion below.
int lstc_attn_bwd(const LstcAttnDesc* d, void* stream) {
    AttnParams p;
    int rc = fill_params(d, p, true);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool bf = d->dtype == LSTC_BF16;
    const int T = (p.S + 31) / 32;
    p.table_rows = (d->index_ld > 0 && d->dtable) ? d->table_rows : 0;
    if (d->index_ld > 0 && d->dtable && d->table_rows <= 0) return LSTC_E_SHAPE;
    if (d->in_pack_cols > 0) {
        rc = fill_packed_inputs(d, p, true);
        if (rc) return rc;
        if (p.dk % 64 || p.table_rows >= 0xFFFF) return LSTC_E_UNSUPPORTED;
    } else if (d->dO_pack_cols > 0) return LSTC_E_UNSUPPORTED;
    const size_t lds = ((size_t)2 * (32 * T) * (32 * T + 1) + (size_t)(NT / 64) * p.table_rows) * sizeof(float);
    if (lds > 160 * 1024) return LSTC_E_RANGE;
    // enough workgroups to fill the chip (>= ~2048) while amortising the bias-table flush over a few sequences
    int npw = (int)(((int64_t)p.N * p.H + 4095) / 4096);
    if (npw < 1) npw = 1;
    if (npw > 8) npw = 8;
    p.table_partials = 0;
    if (p.table_rows > 0 && d->dtable_chunks > 0) {          // the caller fixes the chunking and gets one partial table per chunk
        npw = (p.N + d->dtable_chunks - 1) / d->dtable_chunks;
        p.table_partials = 1;
    }
    if (d->in_pack_cols > 0 && !p.table_partials) {      // packed-input kernel: as its forward (measured: S = 81 1.72 / 1.50 / 1.38 / 1.33 ms at 2 / 4 / 8 / 16)
        npw = (int)(((int64_t)p.N * p.H) / (T == 1 ? 8192 : 1024));
        npw = npw < 1 ? 1 : (npw > 16 ? 16 : npw);
    }
    p.n_per_wg = npw;
    const int chunks_ = (p.N + npw - 1) / npw;
    dim3 grid = ATTN_GRID(chunks_, p.H);
    if (p.table_partials && chunks_ != d->dtable_chunks) return LSTC_E_SHAPE;
    // second-generation kernel (LDS-DMA staged dP, register-resident B rows): d_k, d_v multiples of 32, aligned operands
    // S = 49: 1.63 vs 2.75 ms, S = 17: 0.53 vs 1.19 ms per LTN / STN layer (interleaved A/B); S = 81 (T = 3): 138 KB of LDS =
    // one workgroup per CU; with 4 waves that was no faster than the first generation (193 spilled registers, 7.3 vs 7.1 ms),
    // the 8-wave instantiation (two waves per SIMD, 12 rows / 3 DMA pieces / <= 2 jobs per wave) runs 4.3 ms
    const bool v2 = T <= 3 && p.vec_qk && p.vec_v && (p.dk % 32) == 0 && (p.dv % 32) == 0 && d->variant != 1;
    if (d->dQ_pack || d->dK_pack || d->dV_pack) {
        // packed bf16 gradients: the staged kernel only, token rows and head columns filling the packs' even tile grid exactly
        const int64_t M = (int64_t)p.N * p.S;
        if (!(d->dQ_pack && d->dK_pack && d->dV_pack) || (!v2 && d->in_pack_cols <= 0) || M % 256 || (p.H * p.dk) % 64 || (p.H * p.dv) % 64 ||
            M * (int64_t)(p.H * (p.dk > p.dv ? p.dk : p.dv)) * 2 > 0x7fffffffLL) return LSTC_E_UNSUPPORTED;
        if (!aligned16(d->dQ_pack) || !aligned16(d->dK_pack) || !aligned16(d->dV_pack)) return LSTC_E_ALIGN;
        p.dQp = d->dQ_pack; p.dKp = d->dK_pack; p.dVp = d->dV_pack;
        p.kbq = p.kbk = (p.H * p.dk) / 32;
        p.kbv = (p.H * p.dv) / 32;
        if (d->pack_cols > 0) {           // one pack of [M, pack_cols]: dQ / dK / dV are column blocks of it
            if (d->pack_cols % 64 || d->dQ_col0 % 32 || d->dK_col0 % 32 || d->dV_col0 % 32 || d->dQ_col0 < 0 || d->dK_col0 < 0 || d->dV_col0 < 0 ||
                d->dQ_col0 + p.H * p.dk > d->pack_cols || d->dK_col0 + p.H * p.dk > d->pack_cols || d->dV_col0 + p.H * p.dv > d->pack_cols ||
                M * (int64_t)d->pack_cols * 2 > 0x7fffffffLL) return LSTC_E_SHAPE;
            p.kbq = p.kbk = p.kbv = d->pack_cols / 32;
            p.tq0 = d->dQ_col0 / 32; p.tk0 = d->dK_col0 / 32; p.tv0 = d->dV_col0 / 32;
        }
    }
    if (d->in_pack_cols > 0) {      // third generation: packed bf16 inputs and outputs
        if (!p.dQp) return LSTC_E_UNSUPPORTED;
        return attn3_bwd_launch(p, T, chunks_, st);
    }
    if (v2) {
        const int SP = 32 * T, NW = T == 3 ? 8 : 4;      // 64 < S <= 96: 138 KB of LDS = one workgroup per CU, so that one runs 8 waves
        const size_t lds2 = ((size_t)((2 * SP * (SP + 1) + 3) & ~3) + (size_t)4 * SP * 32 + (size_t)NW * p.table_rows) * sizeof(float);
        if (lds2 <= 160 * 1024) {
#define LSTC_BWD2(TT, BB, WW)                                                              \
    do {                                                                                   \
        static LstcDevOnce once2;                                                          \
        const int dev2_ = once2.begin();                                                   \
        if (dev2_ >= 0) { set_lds(attn_bwd2_kernel<TT, BB, WW>, 160 * 1024); once2.end(dev2_); } \
        hipLaunchKernelGGL((attn_bwd2_kernel<TT, BB, WW>), grid, 64 * WW, lds2, st, p);    \
    } while (0)
            if (T == 1) { if (bf) LSTC_BWD2(1, true, 4); else LSTC_BWD2(1, false, 4); }
            else if (T == 2) { if (bf) LSTC_BWD2(2, true, 4); else LSTC_BWD2(2, false, 4); }
            else { if (bf) LSTC_BWD2(3, true, 8); else LSTC_BWD2(3, false, 8); }
#undef LSTC_BWD2
            return lstc_launch_status();
        }
    }
    if (p.dQp) return LSTC_E_UNSUPPORTED;
#define LSTC_BWD(TT)                                                         \
    do {                                                                     \
        static LstcDevOnce once;                                             \
        const int dev_ = once.begin();                                       \
        if (dev_ >= 0) { set_lds(attn_bwd_kernel<TT>, 160 * 1024); once.end(dev_); } \
        hipLaunchKernelGGL(attn_bwd_kernel<TT>, grid, NT, lds, st, p);       \
    } while (0)
    switch (T) {
        case 1: LSTC_BWD(1); break;
        case 2: LSTC_BWD(2); break;
        case 3: LSTC_BWD(3); break;
        default: LSTC_BWD(4); break;
    }
#undef LSTC_BWD
    return lstc_launch_status();
}

}  // extern "C"

// =====================================================================================================
// CLS-query attention for the LAST encoder layer.  Only enc_output[:, 0, :] is consumed downstream
// (Train/temporal_transformer_shanghaitech.py:123), so the last layer needs queries for token 0 only while its
// keys/values still span every token.  Row 0 carries no relative bias (the reference adds the
// bias to attn[:, :, 1:, 1:] only).  One wave per (sequence, head); lanes own features (coalesced 1-KB K/V rows),
// the S scores live in LDS.  HBM-bound: K and V are streamed once forward, K, V, dK, dV once backward.
// Dropout indices are those of row 0 of the full [N,H,S,S] tensor, so this path reproduces the full kernel's
// CLS output bit for bit, dropout included.
// =====================================================================================================
namespace {

struct ClsParams {
    const float *Q, *K, *V;      // Q [N, ldq] (one row per sequence), K/V [N*S, ld]
    float* O;                    // [N, ldo]
    float* probs;                // [N, H, S]
    const float* dO;
    float *dQ, *dK, *dV;
    int N, S, H, dk, dv, ldq, ldk, ldv, ldo;
    float scale;
    DropKey dkey;
    int has_drop;
};

constexpr int CLS_MAXS = 128;

__global__ void __launch_bounds__(NT) attn_cls_fwd_kernel(const ClsParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    __shared__ float sc[NT / 64][CLS_MAXS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * (NT / 64) + wave;
    if (pair >= p.N * p.H) return;
    const int n = pair / p.H, h = pair % p.H, S = p.S;
    const float* q = p.Q + (size_t)n * p.ldq + (size_t)h * p.dk;
    const float* Kb = p.K + (size_t)n * S * p.ldk + (size_t)h * p.dk;
    const float* Vb = p.V + (size_t)n * S * p.ldv + (size_t)h * p.dv;
    float* s = sc[wave];
    for (int j = 0; j < S; ++j) {
        float a = 0.f;
        for (int c = lane; c < p.dk; c += 64) a += (q[c] * p.scale) * Kb[(size_t)j * p.ldk + c];
        a = wave_sum(a);
        if (lane == 0) s[j] = a;
    }
    __builtin_amdgcn_wave_barrier();
    float v[2], m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = lane + 64 * jj;
        v[jj] = j < S ? s[j] : -INFINITY;
        m = fmaxf(m, v[jj]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        v[jj] = (lane + 64 * jj < S) ? expf(v[jj] - m) : 0.f;
        sum += v[jj];
    }
    sum = wave_sum(sum);
    const uint32_t flat0 = ((uint32_t)n * p.H + h) * (uint32_t)(S * S);       // row 0 of the full tensor
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = lane + 64 * jj;
        if (j < S) {
            float pv = v[jj] / sum;
            p.probs[((size_t)n * p.H + h) * S + j] = pv;
            if (p.has_drop) pv = drop_keep(flat0 + (uint32_t)j, dkn) ? pv * dkn.scale : 0.f;
            s[j] = pv;
        }
    }
    __builtin_amdgcn_wave_barrier();
    float* o = p.O + (size_t)n * p.ldo + (size_t)h * p.dv;
    for (int c = lane; c < p.dv; c += 64) {
        float a = 0.f;
        for (int j = 0; j < S; ++j) a += s[j] * Vb[(size_t)j * p.ldv + c];
        o[c] = a;
    }
}

__global__ void __launch_bounds__(NT) attn_cls_bwd_kernel(const ClsParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    __shared__ float sp[NT / 64][CLS_MAXS];      // dropped probabilities
    __shared__ float sd[NT / 64][CLS_MAXS];      // d(logit)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * (NT / 64) + wave;
    if (pair >= p.N * p.H) return;
    const int n = pair / p.H, h = pair % p.H, S = p.S;
    const float* q = p.Q + (size_t)n * p.ldq + (size_t)h * p.dk;
    const float* Kb = p.K + (size_t)n * S * p.ldk + (size_t)h * p.dk;
    const float* Vb = p.V + (size_t)n * S * p.ldv + (size_t)h * p.dv;
    const float* dO = p.dO + (size_t)n * p.ldo + (size_t)h * p.dv;
    float* dKb = p.dK + (size_t)n * S * p.ldk + (size_t)h * p.dk;
    float* dVb = p.dV + (size_t)n * S * p.ldv + (size_t)h * p.dv;
    float* pd = sp[wave];
    float* ds = sd[wave];
    // dP~_j = dO . V_j
    for (int j = 0; j < S; ++j) {
        float a = 0.f;
        for (int c = lane; c < p.dv; c += 64) a += dO[c] * Vb[(size_t)j * p.ldv + c];
        a = wave_sum(a);
        if (lane == 0) ds[j] = a;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t flat0 = ((uint32_t)n * p.H + h) * (uint32_t)(S * S);
    float pv[2], dp[2], keep[2], rs = 0.f;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = lane + 64 * jj;
        pv[jj] = dp[jj] = keep[jj] = 0.f;
        if (j < S) {
            pv[jj] = p.probs[((size_t)n * p.H + h) * S + j];
            keep[jj] = p.has_drop ? (drop_keep(flat0 + (uint32_t)j, dkn) ? dkn.scale : 0.f) : 1.f;
            dp[jj] = ds[j] * keep[jj];
            rs += dp[jj] * pv[jj];
        }
    }
    rs = wave_sum(rs);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = lane + 64 * jj;
        if (j < S) {
            ds[j] = pv[jj] * (dp[jj] - rs);
            pd[j] = pv[jj] * keep[jj];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // dV_j = Pd_j dO ; dK_j = scale dA_j q ; dq = scale sum_j dA_j K_j
    for (int c = lane; c < p.dv; c += 64) {
        const float g = dO[c];
        for (int j = 0; j < S; ++j) dVb[(size_t)j * p.ldv + c] = pd[j] * g;
    }
    float* dq = p.dQ + (size_t)n * p.ldq + (size_t)h * p.dk;
    for (int c = lane; c < p.dk; c += 64) {
        const float qs = q[c] * p.scale;
        float a = 0.f;
        for (int j = 0; j < S; ++j) {
            a += ds[j] * Kb[(size_t)j * p.ldk + c];
            dKb[(size_t)j * p.ldk + c] = ds[j] * qs;
        }
        dq[c] = a * p.scale;
    }
}

int fill_cls(const LstcAttnDesc* d, ClsParams& p, bool bwd) {
    if (!d) return LSTC_E_NULL;
    if (d->dtype != LSTC_F32) return LSTC_E_UNSUPPORTED;
    if (!d->Q || !d->K || !d->V || !d->probs) return LSTC_E_NULL;
    if (!bwd && !d->O && !d->O_pack) return LSTC_E_NULL;
    const bool gpk = bwd && d->dQ_pack && d->dK_pack && d->dV_pack;
    if (bwd && (!d->dO || (!gpk && (!d->dQ || !d->dK || !d->dV)))) return LSTC_E_NULL;
    if (d->N <= 0 || d->S < 1 || d->H <= 0 || d->dk <= 0 || d->dv <= 0) return LSTC_E_SHAPE;
    if (d->S > CLS_MAXS) return LSTC_E_RANGE;
    if (d->ldq < d->H * d->dk || d->ldk < d->H * d->dk || d->ldv < d->H * d->dv || d->ldo < d->H * d->dv) return LSTC_E_SHAPE;
    if ((uint64_t)d->N * d->H * d->S * d->S > 0xffffffffull) return LSTC_E_RANGE;
    p.Q = (const float*)d->Q; p.K = (const float*)d->K; p.V = (const float*)d->V; p.O = (float*)d->O; p.probs = d->probs;
    p.dO = (const float*)d->dO; p.dQ = (float*)d->dQ; p.dK = (float*)d->dK; p.dV = (float*)d->dV;
    p.N = d->N; p.S = d->S; p.H = d->H; p.dk = d->dk; p.dv = d->dv;
    p.ldq = d->ldq; p.ldk = d->ldk; p.ldv = d->ldv; p.ldo = d->ldo;
    p.scale = d->scale;
    p.has_drop = d->dropout_p > 0.f;
    p.dkey = make_drop_key(d->dropout_p, d->dropout_seed);
    return 0;
}

}  // namespace

extern "C" {

int lstc_attn_cls_fwd(const LstcAttnDesc* d, void* stream) {
    ClsParams p;
    int rc = fill_cls(d, p, false);
    if (rc) return rc;
    const int pairs = p.N * p.H;
    hipLaunchKernelGGL(attn_cls_fwd_kernel, (pairs + NT / 64 - 1) / (NT / 64), NT, 0, (hipStream_t)stream, p);
    return lstc_launch_status();
}

int lstc_attn_cls_bwd(const LstcAttnDesc* d, void* stream) {
    ClsParams p;
    int rc = fill_cls(d, p, true);
    if (rc) return rc;
    const int pairs = p.N * p.H;
    hipLaunchKernelGGL(attn_cls_bwd_kernel, (pairs + NT / 64 - 1) / (NT / 64), NT, 0, (hipStream_t)stream, p);
    return lstc_launch_status();
}

}  // extern "C"

// =====================================================================================================
// Re-associated CLS attention for the last encoder layer ("assoc" path).
// With ONE query per (sequence, head) the key/value projections never have to be materialised:
//   score[n,h,j] = (q[n,h]*scale) . (Wk_h x[n,j]) = u[n,h] . x[n,j],        u[n,h] = (q[n,h]*scale) Wk_h   in R^d
//   o[n,h]       = sum_j p[n,h,j] (Wv_h x[n,j])  = Wv_h xbar[n,h],           xbar[n,h] = sum_j p[n,h,j] x[n,j]
// so the two [N*S, d] x [d, H*dk] projection GEMMs of the layer (and their four backward GEMMs, 5 TFLOP of the 43
// per LTN step) are replaced by per-head [N, dk] x [dk, d] products (batched lstc_gemm) plus the three HBM-bound
// streaming kernels below, each reading or writing X = [N, S, d] exactly once:
//   cls_dot   : out[n,h,j] = sum_c U[n,h,c] X[n,j,c]      (+ softmax / dropout, or softmax backward, fused)
//   cls_wsum  : Y[n,h,c]   = sum_j W[n,h,j] X[n,j,c]
//   cls_outer : dX[n,j,c]  = sum_h W1[n,h,j] U1[n,h,c] + W2[n,h,j] U2[n,h,c]
// One workgroup per sequence; U / W of the sequence sit in LDS, X streams through registers.
// =====================================================================================================
namespace {

constexpr int ASSOC_MAXS = 128;

// The per-(sequence, head) rows of dot products `sc` [H][S] (LDS) -> out / probs; modes of cls_dot_kernel.  One wave per head.
__device__ inline void cls_dot_tail(float* sc, float* __restrict__ out, float* __restrict__ probs, int n, int S, int H, int mode,
                                    const DropKey& dkey, int has_drop, int wave, int lane) {
    for (int h = wave; h < H; h += NT / 64) {
        float* row = sc + h * S;
        float* orow = out + ((size_t)n * H + h) * S;
        if (mode == 0) {
            for (int j = lane; j < S; j += 64) orow[j] = row[j];
            continue;
        }
        float* prow = probs + ((size_t)n * H + h) * S;
        const uint32_t flat0 = ((uint32_t)n * H + h) * (uint32_t)(S * S);     // row 0 of the full [N,H,S,S] tensor
        if (mode == 1) {
            float v[2], m = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = lane + 64 * jj;
                v[jj] = j < S ? row[j] : -INFINITY;
                m = fmaxf(m, v[jj]);
            }
            m = wave_max(m);
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                v[jj] = (lane + 64 * jj < S) ? expf(v[jj] - m) : 0.f;
                s += v[jj];
            }
            s = wave_sum(s);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = lane + 64 * jj;
                if (j < S) {
                    float pv = v[jj] / s;
                    prow[j] = pv;
                    if (has_drop) pv = drop_keep(flat0 + (uint32_t)j, dkey) ? pv * dkey.scale : 0.f;
                    orow[j] = pv;
                }
            }
        } else {
            float pv[2], dp[2], rs = 0.f;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = lane + 64 * jj;
                pv[jj] = dp[jj] = 0.f;
                if (j < S) {
                    pv[jj] = prow[j];
                    const float keep = has_drop ? (drop_keep(flat0 + (uint32_t)j, dkey) ? dkey.scale : 0.f) : 1.f;
                    dp[jj] = row[j] * keep;
                    rs += dp[jj] * pv[jj];
                }
            }
            rs = wave_sum(rs);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = lane + 64 * jj;
                if (j < S) orow[j] = pv[jj] * (dp[jj] - rs);
            }
        }
    }
}

// mode 0: raw dot products; 1: softmax over j, probs -> `probs`, dropout(probs) -> `out`;
// mode 2: `out` = P * (dP - sum_j dP P), dP = dot * keep  (softmax + dropout backward; `probs` is an input)
template <int HT>
__global__ void __launch_bounds__(NT) cls_dot_kernel(const float* __restrict__ U, const float* __restrict__ X,
                                                      float* __restrict__ out, float* __restrict__ probs, int S, int H, int d,
                                                      int mode, DropKey dkey, int has_drop) {
    dkey = drop_key_now(dkey);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ul = sm;                   // [H][d]
    float* sc = sm + (size_t)H * d;   // [H][S]
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Un = U + (size_t)n * H * d;
    const float* Xn = X + (size_t)n * S * d;
    const int d4 = d >> 2;
    for (int i = threadIdx.x; i < H * d4; i += NT) reinterpret_cast<float4*>(Ul)[i] = reinterpret_cast<const float4*>(Un)[i];
    __syncthreads();
    // a wave takes two rows at a time and requests all of their float4 pieces (up to 16 KB in flight per wave) before the first
    // product: the kernel is a stream over X, and the head vectors read from LDS serve both rows
    constexpr int XC = 8, NW = NT / 64;
    for (int j = wave; j < S; j += 2 * NW) {
        const int j1 = j + NW;
        const bool two = j1 < S;
        const float4* x0p = reinterpret_cast<const float4*>(Xn + (size_t)j * d);
        const float4* x1p = reinterpret_cast<const float4*>(Xn + (size_t)(two ? j1 : j) * d);
        float part0[HT], part1[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) part0[h] = part1[h] = 0.f;
        for (int c0 = lane; c0 < d4; c0 += 64 * XC) {
            float4 x0[XC], x1[XC];
#pragma unroll
            for (int i = 0; i < XC; ++i) {
                const int c = min(c0 + 64 * i, d4 - 1);
                x0[i] = x0p[c];
                x1[i] = x1p[c];
            }
#pragma unroll
            for (int i = 0; i < XC; ++i) {
                const int c = c0 + 64 * i;
                if (c < d4) {
#pragma unroll
                    for (int h = 0; h < HT; ++h)
                        if (h < H) {
                            const float4 u = reinterpret_cast<const float4*>(Ul + (size_t)h * d)[c];
                            part0[h] += (x0[i].x * u.x + x0[i].y * u.y) + (x0[i].z * u.z + x0[i].w * u.w);
                            part1[h] += (x1[i].x * u.x + x1[i].y * u.y) + (x1[i].z * u.z + x1[i].w * u.w);
                        }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < HT; ++h)
            if (h < H) {
                const float v0 = wave_sum(part0[h]), v1 = wave_sum(part1[h]);
                if (lane == 0) {
                    sc[h * S + j] = v0;
                    if (two) sc[h * S + j1] = v1;
                }
            }
    }
    __syncthreads();
    cls_dot_tail(sc, out, probs, n, S, H, mode, dkey, has_drop, wave, lane);
}

template <int HT>
__global__ void __launch_bounds__(NT) cls_wsum_kernel(const float* __restrict__ W, const float* __restrict__ X,
                                                       float* __restrict__ Y, int S, int H, int d) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // W[n]: [H][S]
    const int n = blockIdx.x;
    const float* Xn = X + (size_t)n * S * d;
    for (int i = threadIdx.x; i < H * S; i += NT) sm[i] = W[(size_t)n * H * S + i];
    __syncthreads();
    const int d4 = d >> 2;
    for (int c = threadIdx.x; c < d4; c += NT) {
        float4 acc[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < S; ++j) {
            const float4 x = reinterpret_cast<const float4*>(Xn + (size_t)j * d)[c];
#pragma unroll
            for (int h = 0; h < HT; ++h)
                if (h < H) {
                    const float w = sm[h * S + j];
                    acc[h].x += w * x.x; acc[h].y += w * x.y; acc[h].z += w * x.z; acc[h].w += w * x.w;
                }
        }
#pragma unroll
        for (int h = 0; h < HT; ++h)
            if (h < H) reinterpret_cast<float4*>(Y + ((size_t)n * H + h) * d)[c] = acc[h];
    }
}

template <int HT>
__global__ void __launch_bounds__(NT) cls_outer_kernel(const float* __restrict__ W1, const float* __restrict__ U1,
                                                        const float* __restrict__ W2, const float* __restrict__ U2,
                                                        float* __restrict__ dX, int S, int H, int d) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // W1[n], W2[n]: 2 x [H][S]
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < H * S; i += NT) {
        sm[i] = W1[(size_t)n * H * S + i];
        sm[H * S + i] = W2[(size_t)n * H * S + i];
    }
    __syncthreads();
    const int d4 = d >> 2;
    float* Dn = dX + (size_t)n * S * d;
    for (int c = threadIdx.x; c < d4; c += NT) {
        float4 u1[HT], u2[HT];
#pragma unroll
        for (int h = 0; h < HT; ++h) {
            u1[h] = u2[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h < H) {
                u1[h] = reinterpret_cast<const float4*>(U1 + ((size_t)n * H + h) * d)[c];
                u2[h] = reinterpret_cast<const float4*>(U2 + ((size_t)n * H + h) * d)[c];
            }
        }
        for (int j = 0; j < S; ++j) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < HT; ++h)
                if (h < H) {
                    const float a = sm[h * S + j], b = sm[H * S + h * S + j];
                    o.x += a * u1[h].x + b * u2[h].x; o.y += a * u1[h].y + b * u2[h].y;
                    o.z += a * u1[h].z + b * u2[h].z; o.w += a * u1[h].w + b * u2[h].w;
                }
            reinterpret_cast<float4*>(Dn + (size_t)j * d)[c] = o;
        }
    }
}

// ---- the same three passes over a PACKED X (lstc_pack1 of the [N*S, d] matrix): the bf16 activation stream hands the last full
// layer's output to the CLS-only layer as a pack, and takes the gradient back as one.  One workgroup per sequence.
//   dot, outer: the f32 kernels above are VALU-bound (eight heads x every column on the vector unit: 310 / 218 us per 100352 x 2048
//     launch where the bytes need 90 - 170), so these two run on v_mfma_f32_16x16x32_bf16: X's 16-B chunks ARE the A fragments (lane
//     = token l & 15, chunk l >> 4 of the row's 64 B: sixteen tokens of a k tile are one contiguous KB), and every f32 operand
//     enters as a bf16 pair hi + lo (hi = RNE(v), lo = RNE(v - hi): 16 mantissa bits; products of bf16 values are exact in the
//     f32 accumulator) - X itself is bf16 already, so the dot products carry one f32-grade operand and one exact one, and dX is
//     rounded to bf16 at the end anyway.
//   wsum: stays on the vector unit (the token contraction would need transposed fragments): a thread owns ONE 8-column group (a 16-B
//     chunk) of the rows of a sequence, G = d / 8 groups (64 per wave), NT / G row phases; the per-head sums live in registers.
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4c __attribute__((ext_vector_type(4)));
typedef float floatx4c __attribute__((ext_vector_type(4)));
constexpr int PK_RB = 8;         // rows requested per thread before the first one is used (wsum)

__device__ inline void widen8(const bf16x8c& h, float (&f)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (float)h[k];
}

// v[0..8) -> bf16 hi and lo parts (v = hi + lo up to 2^-17 |v|)
__device__ inline void split8(const float (&v)[8], bf16x8c& hi, bf16x8c& lo) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        hi[k] = (__bf16)v[k];
        lo[k] = (__bf16)(v[k] - (float)hi[k]);
    }
}

// out[n, h, j] = sum_c U[n, h, c] X[n, j, c]: D[token, head] = X[token, k] U^T[k, head]; RBT blocks of 16 tokens, wave w takes the
// k tiles w, w + 4, ...; the four waves' partial sums are added in wave order.
template <int RBT>
__global__ void __launch_bounds__(NT) cls_dot_pk_kernel(const float* __restrict__ U, const __bf16* __restrict__ Xp,
                                                         float* __restrict__ out, float* __restrict__ probs, int S, int H, int d,
                                                         int KBp, int mode, DropKey dkey, int has_drop) {
    dkey = drop_key_now(dkey);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sc = sm;                       // [H][S]
    float* scw = sm + H * S;              // [NT / 64][H][S]
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = lane & 15, kg = lane >> 4;
    const float* Un = U + ((size_t)n * H + min(hl, H - 1)) * d + kg * 8;
    const bool live = hl < H;
    const int64_t row0 = (int64_t)n * S;
    floatx4c acc[RBT];
#pragma unroll
    for (int rb = 0; rb < RBT; ++rb) acc[rb] = floatx4c{0.f, 0.f, 0.f, 0.f};
    const int nkt = d >> 5;
    for (int kt = wave; kt < nkt; kt += NT / 64) {
        bf16x8c a[RBT];
#pragma unroll
        for (int rb = 0; rb < RBT; ++rb)
            a[rb] = *reinterpret_cast<const bf16x8c*>(Xp + p1_offset(row0 + min(rb * 16 + hl, S - 1), kt * 32 + kg * 8, KBp));
        const float4 u0 = *reinterpret_cast<const float4*>(Un + kt * 32), u1 = *reinterpret_cast<const float4*>(Un + kt * 32 + 4);
        float uf[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        if (!live) {
#pragma unroll
            for (int k = 0; k < 8; ++k) uf[k] = 0.f;
        }
        bf16x8c bh, bl;
        split8(uf, bh, bl);
#pragma unroll
        for (int rb = 0; rb < RBT; ++rb) {
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], bh, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], bl, acc[rb], 0, 0, 0);
        }
    }
    if (live) {                           // D: column (head) l & 15, rows (tokens) 4 (l >> 4) + r of the block
#pragma unroll
        for (int rb = 0; rb < RBT; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = rb * 16 + kg * 4 + r;
                if (j < S) scw[(wave * H + hl) * S + j] = acc[rb][r];
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H * S; i += NT) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s += scw[w * H * S + i];
        sc[i] = s;
    }
    __syncthreads();
    cls_dot_tail(sc, out, probs, n, S, H, mode, dkey, has_drop, wave, lane);
}

template <int HT>
__global__ void __launch_bounds__(NT) cls_wsum_pk_kernel(const float* __restrict__ W, const __bf16* __restrict__ Xp,
                                                          float* __restrict__ Y, int S, int H, int d, int KBp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Wl = sm;                                         // [S][HT]: a row's weights are one or two 16-B broadcast reads
    float* comb = sm + ((S * HT + 3) & ~3);                 // [RP - 1][HT * 8][G]: the other row phases' sums
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int WPR = d >> 9, RP = (NT / 64) / WPR, G = d >> 3;
    const int cg = (wave % WPR) * 64 + lane, rp = wave / WPR;
    for (int i = threadIdx.x; i < S * HT; i += NT) {
        const int j = i / HT, h = i - j * HT;
        Wl[i] = h < H ? W[((size_t)n * H + h) * S + j] : 0.f;
    }
    __syncthreads();
    float acc[HT][8];
#pragma unroll
    for (int h = 0; h < HT; ++h)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[h][k] = 0.f;
    const int64_t row0 = (int64_t)n * S;
    for (int j0 = rp; j0 < S; j0 += RP * PK_RB) {
        bf16x8c x[PK_RB];
#pragma unroll
        for (int b = 0; b < PK_RB; ++b)
            x[b] = *reinterpret_cast<const bf16x8c*>(Xp + p1_offset(row0 + min(j0 + b * RP, S - 1), 8 * cg, KBp));
#pragma unroll
        for (int b = 0; b < PK_RB; ++b) {
            const int j = j0 + b * RP;
            if (j < S) {
                float xf[8];
                widen8(x[b], xf);
#pragma unroll
                for (int h = 0; h < HT; ++h) {
                    const float w = Wl[j * HT + h];
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[h][k] = fmaf(w, xf[k], acc[h][k]);     // the kernel's time is these FMAs
                }
            }
        }
    }
    if (RP > 1) {                                           // block-uniform
        if (rp > 0) {
#pragma unroll
            for (int h = 0; h < HT; ++h)
#pragma unroll
                for (int k = 0; k < 8; ++k) comb[((size_t)(rp - 1) * HT * 8 + h * 8 + k) * G + cg] = acc[h][k];
        }
        __syncthreads();
        if (rp == 0)
            for (int r = 1; r < RP; ++r)
#pragma unroll
                for (int h = 0; h < HT; ++h)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[h][k] += comb[((size_t)(r - 1) * HT * 8 + h * 8 + k) * G + cg];
    }
    if (rp == 0) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
            if (h < H) {
                float4* q = reinterpret_cast<float4*>(Y + ((size_t)n * H + h) * d + 8 * cg);
                q[0] = make_float4(acc[h][0], acc[h][1], acc[h][2], acc[h][3]);
                q[1] = make_float4(acc[h][4], acc[h][5], acc[h][6], acc[h][7]);
            }
    }
}

// dX[n, j, c] = sum_h W1[n,h,j] U1[n,h,c] + W2[n,h,j] U2[n,h,c] (+ add0[n, c] on row 0: the CLS row's own terms dQ Wq and the
// residual gradient, joined before the one rounding), written as a pack.  Transposed product D'[c, j] = sum_k Ucat[k, c] Wcat[j, k]
// over the 16 (head, operand) pairs, K = 32 per MFMA:
//     MFMA 1   k 0-7: U1hi x W1hi   8-15: U2hi x W2hi   16-23: U1lo x W1hi   24-31: U2lo x W2hi
//     MFMA 2   k 0-7: U1hi x W1lo   8-15: U2hi x W2lo   16-31: x 0                  (the lo x lo terms, 2^-16 of a product, are dropped)
// so the A fragment (lane = column l & 15 of a 16-column half tile, k group l >> 4) is the same in both, and a lane ends with four
// consecutive columns of one token per half; the halves' columns are interleaved so that the two quartets are one 16-B chunk, and a
// store instruction completes sixteen tokens' 64-B lines of the k tile: one contiguous KB.
template <int RBT>
__global__ void __launch_bounds__(NT) cls_outer_pk_kernel(const float* __restrict__ W1, const float* __restrict__ U1,
                                                           const float* __restrict__ W2, const float* __restrict__ U2,
                                                           const float* __restrict__ add0, __bf16* __restrict__ dXp, int S, int H,
                                                           int d, int KBp) {
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hl = lane & 15, kg = lane >> 4;
    // B fragments (tokens): W1 for even k groups, W2 for odd ones; heads h < H, zero beyond
    const float* Wsel = ((kg & 1) ? W2 : W1) + (size_t)n * H * S;
    bf16x8c b1[RBT], b2[RBT];
#pragma unroll
    for (int rb = 0; rb < RBT; ++rb) {
        const int j = min(rb * 16 + hl, S - 1);
        float wf[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) wf[h] = h < H ? Wsel[h * S + j] : 0.f;
        bf16x8c hi, lo;
        split8(wf, hi, lo);
        b1[rb] = hi;
        if (kg >= 2) {
#pragma unroll
            for (int h = 0; h < 8; ++h) lo[h] = (__bf16)0.f;
        }
        b2[rb] = lo;
    }
    const float* Usel = ((kg & 1) ? U2 : U1) + (size_t)n * H * d;
    const int64_t row0 = (int64_t)n * S;
    // a wave's unit = one k tile (32 columns) as two MFMA halves; row i of half t stands for column 8 (i >> 2) + 4 t + (i & 3) of
    // the tile, so the lane's four D rows of half 0 and of half 1 are the EIGHT consecutive columns 8 (l >> 4) ... + 7: one 16-B chunk
    const int ccol = (hl >> 2) * 8 + (hl & 3);
    for (int kt = wave; kt < (d >> 5); kt += NT / 64) {
        const int c0 = kt * 32;
        bf16x8c a[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float uf[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) uf[h] = h < H ? Usel[(size_t)h * d + c0 + ccol + 4 * half] : 0.f;
            bf16x8c hi, lo;
            split8(uf, hi, lo);
            a[half] = kg >= 2 ? lo : hi;
        }
        float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
        if (add0 && hl == 0) {
            const float4* q = reinterpret_cast<const float4*>(add0 + (size_t)n * d + c0 + kg * 8);
            e0 = q[0]; e1 = q[1];
        }
#pragma unroll
        for (int rb = 0; rb < RBT; ++rb) {
            floatx4c o0 = floatx4c{0.f, 0.f, 0.f, 0.f}, o1 = o0;
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b1[rb], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b1[rb], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b2[rb], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b2[rb], o1, 0, 0, 0);
            if (rb == 0) {                                    // token 0 = lanes hl == 0
                o0[0] += e0.x; o0[1] += e0.y; o0[2] += e0.z; o0[3] += e0.w;
                o1[0] += e1.x; o1[1] += e1.y; o1[2] += e1.z; o1[3] += e1.w;
            }
            const int j = rb * 16 + hl;
            if (j < S) {
                bf16x8c hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { hv[r] = (__bf16)o0[r]; hv[4 + r] = (__bf16)o1[r]; }
                *reinterpret_cast<bf16x8c*>(dXp + p1_offset(row0 + j, c0 + kg * 8, KBp)) = hv;
            }
        }
    }
}

int assoc_pk_check(const void* a, const void* xp, const void* c, int64_t N, int S, int H, int d) {
    if (!a || !xp || !c) return LSTC_E_NULL;
    if (N <= 0 || S < 1 || H <= 0 || d <= 0) return LSTC_E_SHAPE;
    if (S > ASSOC_MAXS) return LSTC_E_RANGE;
    // whole 256-row tiles (the pack's grid is exact, as the stream's other kernels need it), a wave per 512 columns, the per-head
    // vectors of 8 heads in registers
    if (H > 8 || (d != 512 && d != 1024 && d != 2048) || (N * S) % 256 != 0) return LSTC_E_UNSUPPORTED;
    if (!aligned16(a) || !aligned16(xp) || !aligned16(c)) return LSTC_E_ALIGN;
    return 0;
}

#define LSTC_RB_DISPATCH(KERN, S, ...)                                           \
    do {                                                                         \
        if ((S) <= 32) hipLaunchKernelGGL(KERN<2>, __VA_ARGS__);                 \
        else if ((S) <= 64) hipLaunchKernelGGL(KERN<4>, __VA_ARGS__);            \
        else if ((S) <= 96) hipLaunchKernelGGL(KERN<6>, __VA_ARGS__);            \
        else hipLaunchKernelGGL(KERN<8>, __VA_ARGS__);                           \
    } while (0)

#define LSTC_H8_DISPATCH(KERN, H, ...)                                           \
    do {                                                                         \
        if ((H) <= 2) hipLaunchKernelGGL(KERN<2>, __VA_ARGS__);                  \
        else if ((H) <= 4) hipLaunchKernelGGL(KERN<4>, __VA_ARGS__);             \
        else hipLaunchKernelGGL(KERN<8>, __VA_ARGS__);                           \
    } while (0)

int assoc_check(const void* a, const void* b, const void* c, int64_t N, int S, int H, int d) {
    if (!a || !b || !c) return LSTC_E_NULL;
    if (N <= 0 || S < 1 || H <= 0 || d <= 0) return LSTC_E_SHAPE;
    if (S > ASSOC_MAXS || H > 16) return LSTC_E_RANGE;
    if (d % 4 != 0 || !aligned16(a) || !aligned16(b) || !aligned16(c)) return LSTC_E_ALIGN;
    return 0;
}

#define LSTC_H_DISPATCH(KERN, H, ...)                                            \
    do {                                                                         \
        if ((H) <= 2) hipLaunchKernelGGL(KERN<2>, __VA_ARGS__);                  \
        else if ((H) <= 4) hipLaunchKernelGGL(KERN<4>, __VA_ARGS__);             \
        else if ((H) <= 8) hipLaunchKernelGGL(KERN<8>, __VA_ARGS__);             \
        else hipLaunchKernelGGL(KERN<16>, __VA_ARGS__);                          \
    } while (0)

}  // namespace

extern "C" {

int lstc_cls_dot(const float* U, const float* X, float* out, float* probs, int64_t N, int32_t S, int32_t H, int32_t d,
                 int32_t mode, float dropout_p, uint64_t seed, void* stream) {
    int rc = assoc_check(U, X, out, N, S, H, d);
    if (rc) return rc;
    if (mode < 0 || mode > 2) return LSTC_E_UNSUPPORTED;
    if (mode != 0 && !probs) return LSTC_E_NULL;
    if ((uint64_t)N * H * S * S > 0xffffffffull) return LSTC_E_RANGE;
    const size_t lds = ((size_t)H * d + (size_t)H * S) * sizeof(float);
    if (lds > 96 * 1024) return LSTC_E_RANGE;
    static LstcDevOnce once;
    const int dev_ = once.begin();
    if (dev_ >= 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cls_dot_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cls_dot_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cls_dot_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cls_dot_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        once.end(dev_);
    }
    const DropKey dk = make_drop_key(dropout_p, seed);
    const int has_drop = dropout_p > 0.f;
    LSTC_H_DISPATCH(cls_dot_kernel, H, dim3((unsigned)N), dim3(NT), lds, (hipStream_t)stream, U, X, out, probs, S, H, d, mode, dk, has_drop);
    return lstc_launch_status();
}

int lstc_cls_wsum(const float* W, const float* X, float* Y, int64_t N, int32_t S, int32_t H, int32_t d, void* stream) {
    int rc = assoc_check(X, Y, X, N, S, H, d);
    if (rc) return rc;
    if (!W) return LSTC_E_NULL;
    const size_t lds = (size_t)H * S * sizeof(float);
    LSTC_H_DISPATCH(cls_wsum_kernel, H, dim3((unsigned)N), dim3(NT), lds, (hipStream_t)stream, W, X, Y, S, H, d);
    return lstc_launch_status();
}

int lstc_cls_outer(const float* W1, const float* U1, const float* W2, const float* U2, float* dX, int64_t N, int32_t S,
                   int32_t H, int32_t d, void* stream) {
    int rc = assoc_check(U1, U2, dX, N, S, H, d);
    if (rc) return rc;
    if (!W1 || !W2) return LSTC_E_NULL;
    const size_t lds = (size_t)2 * H * S * sizeof(float);
    LSTC_H_DISPATCH(cls_outer_kernel, H, dim3((unsigned)N), dim3(NT), lds, (hipStream_t)stream, W1, U1, W2, U2, dX, S, H, d);
    return lstc_launch_status();
}

int lstc_cls_dot_pack(const float* U, const void* X_pack, float* out, float* probs, int64_t N, int32_t S, int32_t H, int32_t d,
                      int32_t mode, float dropout_p, uint64_t seed, void* stream) {
    int rc = assoc_pk_check(U, X_pack, out, N, S, H, d);
    if (rc) return rc;
    if (mode < 0 || mode > 2) return LSTC_E_UNSUPPORTED;
    if (mode != 0 && !probs) return LSTC_E_NULL;
    if ((uint64_t)N * H * S * S > 0xffffffffull) return LSTC_E_RANGE;
    const size_t lds = ((size_t)H * S + (size_t)(NT / 64) * H * S) * sizeof(float);
    const DropKey dk = make_drop_key(dropout_p, seed);
    const int has_drop = dropout_p > 0.f;
    LSTC_RB_DISPATCH(cls_dot_pk_kernel, S, dim3((unsigned)N), dim3(NT), lds, (hipStream_t)stream, U, (const __bf16*)X_pack, out, probs,
                     S, H, d, d / 32, mode, dk, has_drop);
    return lstc_launch_status();
}

int lstc_cls_wsum_pack(const float* W, const void* X_pack, float* Y, int64_t N, int32_t S, int32_t H, int32_t d, void* stream) {
    int rc = assoc_pk_check(W, X_pack, Y, N, S, H, d);
    if (rc) return rc;
    const int HT = H <= 2 ? 2 : H <= 4 ? 4 : 8;
    const int RP = (NT / 64) / (d >> 9);
    const size_t lds = ((size_t)((S * HT + 3) & ~3) + (size_t)(RP - 1) * HT * 8 * (d >> 3)) * sizeof(float);
    LSTC_H8_DISPATCH(cls_wsum_pk_kernel, H, dim3((unsigned)N), dim3(NT), lds, (hipStream_t)stream, W, (const __bf16*)X_pack, Y, S, H, d,
                     d / 32);
    return lstc_launch_status();
}

int lstc_cls_outer_pack(const float* W1, const float* U1, const float* W2, const float* U2, const float* add0, void* dX_pack,
                        int64_t N, int32_t S, int32_t H, int32_t d, void* stream) {
    int rc = assoc_pk_check(U1, dX_pack, U2, N, S, H, d);
    if (rc) return rc;
    if (!W1 || !W2) return LSTC_E_NULL;
    if (add0 && !aligned16(add0)) return LSTC_E_ALIGN;
    LSTC_RB_DISPATCH(cls_outer_pk_kernel, S, dim3((unsigned)N), dim3(NT), 0, (hipStream_t)stream, W1, U1, W2, U2, add0,
                     (__bf16*)dX_pack, S, H, d, d / 32);
    return lstc_launch_status();
}

}  // extern "C"
