// Attention core of the bf16 mode on PACKED operands (third generation; LstcAttnDesc.in_pack_cols > 0), gfx950.
// models/MultiHeadAttention.py:103-122 and its autograd, as csrc/attention.hip - see the block comment below for the design.
#include "attention_common.h"

namespace {
using namespace lstc_attn;

// =====================================================================================================
// Third generation (bf16 mode, PACKED operands; S <= 96, d_k and d_v multiples of 64).  Q | K | V (and dO) arrive as the
// lstc_pack1 buffers the projection GEMMs write with LSTC_EPI_OUT_PACK (2 B per element, 64-B rows of 32 features, 16-B chunk
// index XOR (row >> 2) & 3), O / dQ | dK | dV leave as packs: no f32 activation of the attention core touches HBM.
//   * T = ceil(S / 32) CONSUMER waves per (sequence, head) plus ONE PRODUCER wave.  Consumer w owns the 32 QUERIES
//     32 w .. 32 w + 31 through the whole item.
//   * Every product is SWAPPED so that the query index sits on the lane: logits^T[j][i] = K Q^T (A = K rows, B = Q rows) leaves
//     lane = i, registers = 16 T keys j -> the softmax is lane-local plus ONE exchange with lane ^ 32, and the dropped
//     probabilities, rounded to bf16 pairwise, ARE the A operand of O = Pd V (MFMA accumulator layout = A-operand layout with a
//     permuted k order; the B operand follows the same order).  No logit / probability tile in LDS at all.
//   * Operands are staged by LDS-DMA exactly as they lie in the packs (16 rows x 64 B per wave instruction).  The swizzle key of
//     a row is that of its GLOBAL row, so readers XOR with ((n S + row) >> 2) & 3.  Feature contractions read fragments with
//     ds_read_b128 (conflict-free for any row offset), token contractions read the SAME image with ds_read_b64_tr_b16 (4
//     consecutive rows x 64 B per half-wave: conflict-free by construction).
//   * One ring of NB slots carries every staged unit of a workgroup - {Q chunk, K chunk} x d_k/32, then {V tile, V tile} x
//     d_v/64 - across the sequences a workgroup walks, so the next item's first chunks are in flight under the current item's
//     P V.  The producer wave issues ALL the DMA and nothing else: its vmcnt counts DMA pieces only, so "unit g has landed" is
//     one s_waitcnt with a known count however many stores the consumers have in flight (first version: every wave issued its
//     share and had to count its own stores into the wait - correct only while the compiler emits exactly the expected memory
//     instructions, and limited to 4 slots by the 6-bit counter).  Per unit: producer waits, ONE barrier (data visible to the
//     consumers, slot g - 1 free), producer refills slot g - 1 with unit g + NB - 1, consumers compute unit g.
//     What the kernels need is BYTES IN FLIGHT: measured bandwidth followed (NB - 1) x slot x workgroups per CU (72 KB: 2.4-2.8
//     TB/s, 96 KB: 4.8 TB/s), so NB is chosen per instantiation to fill the LDS the resident workgroups can share.
// =====================================================================================================
typedef short a3_s4 __attribute__((ext_vector_type(4)));
typedef short a3_s8 __attribute__((ext_vector_type(8)));
typedef uint32_t a3_u4 __attribute__((ext_vector_type(4)));
typedef float a3_f4 __attribute__((ext_vector_type(4)));
typedef a3_s4 __attribute__((address_space(3))) * a3_lds4;

#ifndef A3_DMA_MOD
#define A3_DMA_MOD ""                    // cache-policy modifier of the ring's LDS-DMA.  " nt" (once-read operands) measured, round 6: bf16 step
                                         // 36.58 / 36.58 / 36.57 vs 36.54 / 36.61 / 36.63 ms - nothing (profiles/r06_attn_nt_ab.txt)
#endif
__device__ __forceinline__ void a3_dma(const __bf16* tile, uint32_t voff, uint32_t lds_bytes) {
    const uint32_t lb = __builtin_amdgcn_readfirstlane(lds_bytes);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" A3_DMA_MOD :: "v"(voff), "s"(tile), "s"(lb) : "memory");
}
// byte offset, inside tile column 0 of a pack with `kb` tiles per row block, of this lane's 16-B piece of global row g
__device__ __forceinline__ uint32_t a3_row_off(uint32_t g, uint32_t kb, uint32_t lane) {
    return ((g >> 7) * kb * 4096u + (g & 127u) * 32u) * 2u + (lane & 3u) * 16u;
}
__device__ __forceinline__ attn_h8 a3_tr(const char* p0, const char* p1) {
    const a3_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a3_lds4)p0);
    const a3_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a3_lds4)p1);
    a3_s8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return __builtin_bit_cast(attn_h8, f);
}
__device__ __forceinline__ attn_h8 a3_row(const char* p) { return *reinterpret_cast<const attn_h8*>(p); }
// A 32 x 32 result tile held TRANSPOSED - lane = token row, registers = columns (r & 3) + 8 (r >> 2) + 4 h2 of the tile - written, rounded
// to bf16, into a pack.  One v_permlane32_swap per dword hands the lower lanes the 16-B chunks 0 and 1 of their row and the upper
// lanes chunks 2 and 3, so a tile leaves as TWO 16-B stores per lane.  (First version: lane = column, eight 4-B stores per tile and
// lane - the same bytes for 4x the address-unit cycles, a quarter of a millisecond per S = 81 backward launch.)
// rowb: byte offset of the lane's row inside tile column 0 of the pack; key: its swizzle key (row >> 2) & 3; valid: row < S.
__device__ __forceinline__ void a3_store_tile(const floatx16& o, float scale, __amdgpu_buffer_rsrc_t rs, uint32_t rowb, uint32_t key,
                                              bool valid, uint32_t tile, int h2) {
    uint32_t d[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            attn_f2 f;
            f[0] = o[4 * g + 2 * k] * scale;
            f[1] = o[4 * g + 2 * k + 1] * scale;
            d[g][k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, attn_h2));
        }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const auto x = __builtin_amdgcn_permlane32_swap(d[0][k], d[2][k], false, false);
        d[0][k] = x[0]; d[2][k] = x[1];
        const auto y = __builtin_amdgcn_permlane32_swap(d[1][k], d[3][k], false, false);
        d[1][k] = y[0]; d[3][k] = y[1];
    }
    const uint32_t base = rowb + tile * 8192u;
    const a3_u4 ca = {d[0][0], d[0][1], d[2][0], d[2][1]}, cb = {d[1][0], d[1][1], d[3][0], d[3][1]};
    __builtin_amdgcn_raw_buffer_store_b128(ca, rs, valid ? base + ((((uint32_t)(2 * h2)) ^ key) << 4) : 0xFFFFFFFFu, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(cb, rs, valid ? base + ((((uint32_t)(2 * h2 + 1)) ^ key) << 4) : 0xFFFFFFFFu, 0, 0);
}
// lgkmcnt(0) + barrier: the wave's LDS reads / writes are done; vmcnt untouched (the ring's DMA and the stores stay in flight)
#define A3_LDS_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define A3_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14))     /* vmcnt(N) */

// One unit of the ring: two [SP rows][64 B] halves, each a column tile of a pack.
struct A3Unit { const __bf16 *b0, *b1; uint32_t kb0, kb1; };

// Producer wave: `total` units, unit k described by unit_of(item, u); P = 4 T DMA pieces per unit.  `extra(u)`: barriers the
// consumers run inside step u beyond the ring's own (the backward's transposition image).
template <int T, int NB, typename UnitOf, typename Extra>
__device__ __forceinline__ void a3_producer(int n_begin, int n_end, int U, int S, UnitOf unit_of, Extra extra) {
    constexpr int SP = 32 * T, HALF = SP * 64, SLOT = 2 * HALF, P = 4 * T;
    static_assert(P * (NB - 2) <= 63, "vmcnt is a 6-bit counter");
    const int lane = threadIdx.x & 63;
    const int total = (n_end - n_begin) * U;
    int is_n = n_begin, is_u = 0, is_slot = 0, issued = 0;
    auto issue_next = [&]() {
        if (issued >= total) return;
        const A3Unit un = unit_of(is_u);
        const uint32_t lb = (uint32_t)(is_slot * SLOT);
#pragma unroll
        for (int pc = 0; pc < 2 * T; ++pc) {
            const uint32_t g = (uint32_t)is_n * (uint32_t)S + (uint32_t)min(16 * pc + (lane >> 2), S - 1);
            a3_dma(un.b0, a3_row_off(g, un.kb0, lane), lb + pc * 1024);
            a3_dma(un.b1, a3_row_off(g, un.kb1, lane), lb + HALF + pc * 1024);
        }
        ++issued;
        if (++is_u == U) { is_u = 0; ++is_n; }
        if (++is_slot == NB) is_slot = 0;
    };
#pragma unroll 1
    for (int k = 0; k < NB - 1; ++k) issue_next();
    int g = 0;
#pragma unroll 1
    for (int n = n_begin; n < n_end; ++n)
#pragma unroll 1
        for (int u = 0; u < U; ++u) {
            // units younger than unit g: min(NB - 2, total - 1 - g), P pieces each
            const int rem = min(NB - 2, total - 1 - g);
            if (rem >= NB - 2) A3_WAIT_VM(P * (NB - 2));
            else if (NB > 3 && rem == NB - 3) A3_WAIT_VM(P * (NB > 3 ? NB - 3 : 0));
            else if (NB > 4 && rem == NB - 4) A3_WAIT_VM(P * (NB > 4 ? NB - 4 : 0));
            else A3_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
            issue_next();
            const int nx = extra(u);
            for (int x = 0; x < nx; ++x) __builtin_amdgcn_s_barrier();
            ++g;
        }
}

template <int T, int NB>
__global__ void __launch_bounds__(64 * (T + 1)) attn_fwd3_kernel(const AttnParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    constexpr int SP = 32 * T, HALF = SP * 64, SLOT = 2 * HALF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const char* ring = reinterpret_cast<const char*>(sm);
    // 1-D grid, head fastest: the workgroups resident together read the SAME token rows (adjacent column tiles of the packs)
    const int h = (int)(blockIdx.x % (unsigned)p.H), chunk = (int)(blockIdx.x / (unsigned)p.H), S = p.S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const int n_begin = chunk * p.n_per_wg, n_end = min(p.N, n_begin + p.n_per_wg);
    const int nq = p.dk >> 5, nvp = p.dv >> 6, U = nq + nvp;
    if (wave == T) {
        const int tq = p.iq0 + ((h * p.dk) >> 5), tk = p.ik0 + ((h * p.dk) >> 5), tv = p.iv0 + ((h * p.dv) >> 5);
        a3_producer<T, NB>(n_begin, n_end, U, S,
            [&](int u) -> A3Unit {
                A3Unit un;
                if (u < nq) {
                    un.b0 = p.Qi + (size_t)(tq + u) * 4096; un.kb0 = (uint32_t)p.kiq;
                    un.b1 = p.Ki + (size_t)(tk + u) * 4096; un.kb1 = (uint32_t)p.kik;
                } else {
                    un.b0 = p.Vi + (size_t)(tv + 2 * (u - nq)) * 4096; un.kb0 = un.kb1 = (uint32_t)p.kiv;
                    un.b1 = un.b0 + 4096;
                }
                return un;
            },
            [](int) { return 0; });
        return;
    }
    const int i = 32 * wave + l31;
    // relative-position bias of this lane's (i, j) pairs: two dependent loads per element, once per workgroup
    const bool has_bias = p.index_ld > 0 && S > 1;
    const __amdgpu_buffer_rsrc_t r_index = __builtin_amdgcn_make_buffer_rsrc(const_cast<int64_t*>(p.index), 0,
        has_bias ? (S - 1) * p.index_ld * 8 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_table = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.table), 0,
        has_bias ? (p.table_rows > 0 ? p.table_rows * p.H * 4 : 0x7fffffff) : 0, 0x00020000);
    float biasr[T][16];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * t + 8 * (r >> 2) + 4 * h2 + (r & 3);
            // buffer loads whose offset is out of range where no bias applies (they return 0): a plain load under a lane-dependent
            // condition is waited for on the spot - 16 T exposed double latencies per workgroup
            const bool pair = i >= 1 && j >= 1 && i < S && j < S;
            const uint32_t ix = __builtin_amdgcn_raw_buffer_load_b32(r_index, pair ? (uint32_t)((i - 1) * p.index_ld + (j - 1)) * 8u : 0xFFFFFFFFu, 0, 0);
            // (the OR keeps the first load's result in use on every lane: a select would let the compiler sink that load into a branch)
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_table, ((ix * (uint32_t)p.H + (uint32_t)h) * 4u) | (pair ? 0u : 0xFFFFFFFFu), 0, 0));
            biasr[t][r] = b + (j < S ? 0.f : -INFINITY);
        }
    // (keys j >= S carry a bias of -inf: their logits, exponentials and probabilities come out as -inf, 0, 0 with no predicate)
    const __amdgpu_buffer_rsrc_t r_Op = __builtin_amdgcn_make_buffer_rsrc(p.Op, 0, (int)0x7fffffff, 0x00020000);
    const int q4 = (lane >> 2) & 3, p4 = lane & 3, gg = (lane >> 4) & 1;
    int cslot = 0;
    floatx16 acc[T];
    attn_h8 pf[T][2];
#pragma unroll 1
    for (int n = n_begin; n < n_end; ++n) {
        const uint32_t r0 = (uint32_t)n * (uint32_t)S;
        const uint32_t keyq = ((r0 + (uint32_t)i) >> 2) & 3u, keyk = ((r0 + (uint32_t)l31) >> 2) & 3u;
        const int offq0 = i * 64 + (int)(((0 + h2) ^ keyq) << 4), offq1 = i * 64 + (int)(((2 + h2) ^ keyq) << 4);
        const int offk0 = l31 * 64 + (int)(((0 + h2) ^ keyk) << 4), offk1 = l31 * 64 + (int)(((2 + h2) ^ keyk) << 4);
        int troff[2];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const uint32_t row = (uint32_t)(4 * h2 + 8 * rd + q4);
            troff[rd] = (int)(row * 64u + ((((uint32_t)(2 * gg + (p4 >> 1))) ^ (((r0 + row) >> 2) & 3u)) << 4) + 8u * (uint32_t)(p4 & 1));
        }
        const uint32_t orow = a3_row_off(r0 + (uint32_t)i, (uint32_t)p.kbo, 0);      // this lane's row of the O pack
#pragma unroll 1
        for (int u = 0; u < U; ++u) {
            A3_LDS_BARRIER();
            const char* slot = ring + cslot * SLOT;
            if (u < nq) {
                if (u == 0) {
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
                }
                const attn_h8 fq0 = a3_row(slot + offq0), fq1 = a3_row(slot + offq1);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const attn_h8 fk0 = a3_row(slot + HALF + t * 2048 + offk0), fk1 = a3_row(slot + HALF + t * 2048 + offk1);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk0, fq0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk1, fq1, acc[t], 0, 0, 0);
                }
                if (u == nq - 1) {
                    // ---- softmax of row i over this lane's 16 T keys and those of lane ^ 32
                    float m = -INFINITY;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float x = acc[t][r] * p.scale + biasr[t][r];
                            acc[t][r] = x;
                            m = fmaxf(m, x);
                        }
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float e = expf(acc[t][r] - m);
                            acc[t][r] = e;
                            sum += e;
                        }
                    sum += __shfl_xor(sum, 32, 64);
                    const float inv = 1.f / sum;
                    // probabilities: 16-B buffer stores (rows >= S, groups beyond the row pitch: dropped by the range check)
                    const __amdgpu_buffer_rsrc_t r_pr = __builtin_amdgcn_make_buffer_rsrc(
                        p.probs + ((size_t)n * p.H + h) * S * p.pld, 0, S * p.pld * 4, 0x00020000);
                    const uint32_t flat_i = ((uint32_t)n * p.H + h) * (uint32_t)(S * S) + (uint32_t)(i * S);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int j0 = 32 * t + 8 * g4 + 4 * h2;
                            a3_f4 pv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pv[e] = acc[t][4 * g4 + e] * inv;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(a3_u4, pv), r_pr,
                                (i < S && j0 < p.pld) ? (uint32_t)(i * p.pld + j0) * 4u : 0xFFFFFFFFu, 0, 0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = pv[e];
                                if (p.has_drop) v = drop_keep(flat_i + (uint32_t)(j0 + e), dkn) ? v * dkn.scale : 0.f;
                                acc[t][4 * g4 + e] = v;
                            }
                        }
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int e = 0; e < 8; ++e) pf[t][s2][e] = (__bf16)acc[t][8 * s2 + e];
                    }
                }
            } else {
                // ---- O[i-tile, two 32-column tiles] = Pd V
                const int ct = 2 * (u - nq);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const char* Vs = slot + hf * HALF;
                    floatx16 o[1];
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[0][r] = 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const attn_h8 fv = a3_tr(Vs + t * 2048 + s2 * 1024 + troff[0], Vs + t * 2048 + s2 * 1024 + troff[1]);
                            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv, pf[t][s2], o[0], 0, 0, 0);      // O^T: lane = query
                        }
                    a3_store_tile(o[0], 1.f, r_Op, orow, keyq, i < S, (uint32_t)(((h * p.dv) >> 5) + ct + hf), h2);
                }
            }
            if (++cslot == NB) cslot = 0;
        }
    }
}

// Third-generation backward.  Same ownership (consumer w = queries 32 w ..), same ring and producer, per item:
//   A   {dO chunk, V chunk} x d_v/32 : dP^T[j][i] = V dO^T on the lane-=-query layout; then in registers
//       dA = P (dP keep - rowsum(dP keep P)), Pd = P keep; the bias-table gradient goes to a per-wave LDS table by ds_add_f32;
//       dA (bf16) is the A operand of dQ as it stands; Pd and dA cross the lanes ONCE each through a [query][32 keys] LDS image
//       with 72-B rows (8-B stores conflict-free, transposed reads 2-way at worst) and come back as the A operands of dV and
//       dK with the KEY on the lane: wave w then owns the 32 keys 32 w .. of those two products.  The image holds ONE key
//       panel (SP x 72 B): panel t is written by every wave and read by wave t, T rounds per matrix;
//   C1  {K tile, K tile} x d_k/64    : dQ[queries of w] = scale dA K       (B operand by transposed reads, accumulator k order)
//   C2  {dO tile, dO tile} x d_v/64  : dV[keys of w]    = Pd^T dO          (natural k order)
//   C3  {Q tile, Q tile} x d_k/64    : dK[keys of w]    = scale dA^T Q
template <int T, int NB>
__global__ void __launch_bounds__(64 * (T + 1)) attn_bwd3_kernel(const AttnParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    constexpr int SP = 32 * T, HALF = SP * 64, SLOT = 2 * HALF;
    constexpr int ILD = 72, IMG = SP * ILD;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    char* ring = reinterpret_cast<char*>(sm);
    char* img = ring + NB * SLOT;
    float* tacc = reinterpret_cast<float*>(img + IMG);
    // 1-D grid, head fastest: the workgroups resident together read the SAME token rows (adjacent column tiles of the packs)
    const int h = (int)(blockIdx.x % (unsigned)p.H), chunk = (int)(blockIdx.x / (unsigned)p.H), S = p.S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const int n_begin = chunk * p.n_per_wg, n_end = min(p.N, n_begin + p.n_per_wg);
    const int nv = p.dv >> 5, nq2 = p.dk >> 6, nv2 = p.dv >> 6;
    const int u_c1 = nv, u_c2 = nv + nq2, u_c3 = nv + nq2 + nv2, U = nv + 2 * nq2 + nv2;
    const bool has_bias = p.index_ld > 0 && p.dtable != nullptr;
    if (has_bias)
        for (int x = threadIdx.x; x < T * p.table_rows; x += 64 * (T + 1)) tacc[x] = 0.f;
    __syncthreads();          // tacc zeroed (before any DMA is in flight)
    if (wave == T) {
        const int tq = p.iq0 + ((h * p.dk) >> 5), tk = p.ik0 + ((h * p.dk) >> 5), tv = p.iv0 + ((h * p.dv) >> 5), tdo = p.ido0 + ((h * p.dv) >> 5);
        a3_producer<T, NB>(n_begin, n_end, U, S,
            [&](int u) -> A3Unit {
                A3Unit un;
                if (u < u_c1) {
                    un.b0 = p.dOi + (size_t)(tdo + u) * 4096; un.kb0 = (uint32_t)p.kido;
                    un.b1 = p.Vi + (size_t)(tv + u) * 4096; un.kb1 = (uint32_t)p.kiv;
                } else if (u < u_c2) {
                    un.b0 = p.Ki + (size_t)(tk + 2 * (u - u_c1)) * 4096; un.kb0 = un.kb1 = (uint32_t)p.kik; un.b1 = un.b0 + 4096;
                } else if (u < u_c3) {
                    un.b0 = p.dOi + (size_t)(tdo + 2 * (u - u_c2)) * 4096; un.kb0 = un.kb1 = (uint32_t)p.kido; un.b1 = un.b0 + 4096;
                } else {
                    un.b0 = p.Qi + (size_t)(tq + 2 * (u - u_c3)) * 4096; un.kb0 = un.kb1 = (uint32_t)p.kiq; un.b1 = un.b0 + 4096;
                }
                return un;
            },
            [&](int u) { return u == u_c1 - 1 ? 4 * T : 0; });
    } else {
    const int i = 32 * wave + l31;
    float* const tw = tacc + wave * p.table_rows;
    // bias-table rows of this lane's (i, j) pairs, two per register (0xFFFF: none)
    const __amdgpu_buffer_rsrc_t r_index = __builtin_amdgcn_make_buffer_rsrc(const_cast<int64_t*>(p.index), 0,
        (has_bias && S > 1) ? (S - 1) * p.index_ld * 8 : 0, 0x00020000);
    uint32_t idxp[T][8];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r2 = 0; r2 < 8; ++r2) {
            uint32_t w2 = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * r2 + e;
                const int j = 32 * t + 8 * (r >> 2) + 4 * h2 + (r & 3);
                const bool pair = has_bias && i >= 1 && j >= 1 && i < S && j < S;      // out-of-range offset reads 0 (see the forward)
                const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(r_index, pair ? (uint32_t)((i - 1) * p.index_ld + (j - 1)) * 8u : 0xFFFFFFFFu, 0, 0)
                                   | (pair ? 0u : 0xFFFFu);
                w2 |= (v & 0xFFFFu) << (16 * e);
            }
            idxp[t][r2] = w2;
        }
    const __amdgpu_buffer_rsrc_t r_dQ = __builtin_amdgcn_make_buffer_rsrc(p.dQp, 0, (int)0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dK = __builtin_amdgcn_make_buffer_rsrc(p.dKp, 0, (int)0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dV = __builtin_amdgcn_make_buffer_rsrc(p.dVp, 0, (int)0x7fffffff, 0x00020000);
    const int q4 = (lane >> 2) & 3, p4 = lane & 3, gg = (lane >> 4) & 1;
    // image addresses: 8-B store of keys 8 g4 + 4 h2 .. + 3 (of the panel's 32) of query i; transposed read of queries
    // 8 h2 + 4 rd + q4 (+ 32 ti + 16 s2), keys 16 gg + 4 p4 ..
    char* const img_w = img + i * ILD + 8 * h2;
    const char* const img_r = img + (8 * h2 + q4) * ILD + 32 * gg + 8 * p4;
    int cslot = 0;
    floatx16 acc[T];
    float prr[T][16];
    attn_h8 dAf[T][2], PdTf[T][2], dATf[T][2];
    // the pair's table row depends on (i, j) only: one LDS atomic per pair and WORKGROUP, not per sequence (first version:
    // 48 ds_add_f32 per lane and item kept the CU's LDS pipe busy for 0.6 ms of a 2.1-ms S = 81 launch)
    float dsum[T][16];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dsum[t][r] = 0.f;
#pragma unroll 1
    for (int n = n_begin; n < n_end; ++n) {
        const uint32_t r0 = (uint32_t)n * (uint32_t)S;
        const uint32_t keyq = ((r0 + (uint32_t)i) >> 2) & 3u, keyk = ((r0 + (uint32_t)l31) >> 2) & 3u;
        const int offq0 = i * 64 + (int)(((0 + h2) ^ keyq) << 4), offq1 = i * 64 + (int)(((2 + h2) ^ keyq) << 4);
        const int offk0 = l31 * 64 + (int)(((0 + h2) ^ keyk) << 4), offk1 = l31 * 64 + (int)(((2 + h2) ^ keyk) << 4);
        int troff[2], trnat[2];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const uint32_t row = (uint32_t)(4 * h2 + 8 * rd + q4), rown = (uint32_t)(8 * h2 + 4 * rd + q4);
            troff[rd] = (int)(row * 64u + ((((uint32_t)(2 * gg + (p4 >> 1))) ^ (((r0 + row) >> 2) & 3u)) << 4) + 8u * (uint32_t)(p4 & 1));
            trnat[rd] = (int)(rown * 64u + ((((uint32_t)(2 * gg + (p4 >> 1))) ^ (((r0 + rown) >> 2) & 3u)) << 4) + 8u * (uint32_t)(p4 & 1));
        }
#pragma unroll 1
        for (int u = 0; u < U; ++u) {
            A3_LDS_BARRIER();
            const char* slot = ring + cslot * SLOT;
            if (u < u_c1) {
                if (u == 0) {
                    // the saved probabilities of this lane's pairs land under phase A: 16-B loads at a clamped address, result
                    // masked (__builtin_amdgcn_raw_buffer_load_b128 of this toolchain loads ONE dword and splats it)
                    const float* pr_item = p.probs + ((size_t)n * p.H + h) * S * p.pld;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int j0 = 32 * t + 8 * g4 + 4 * h2;
                            const bool ok = i < S && j0 < p.pld;
                            const a3_u4 v = *reinterpret_cast<const a3_u4*>(pr_item + (ok ? i * p.pld + j0 : 0));
                            const uint32_t mask = ok ? 0xFFFFFFFFu : 0u;
#pragma unroll
                            for (int e = 0; e < 4; ++e) prr[t][4 * g4 + e] = __builtin_bit_cast(float, v[e] & mask);     // padding columns hold the forward's zeros
                        }
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
                }
                const attn_h8 fo0 = a3_row(slot + offq0), fo1 = a3_row(slot + offq1);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const attn_h8 fv0 = a3_row(slot + HALF + t * 2048 + offk0), fv1 = a3_row(slot + HALF + t * 2048 + offk1);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv0, fo0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv1, fo1, acc[t], 0, 0, 0);
                }
                if (u == u_c1 - 1) {
                    typedef __bf16 h4v __attribute__((ext_vector_type(4)));
                    const uint32_t flat_i = ((uint32_t)n * p.H + h) * (uint32_t)(S * S) + (uint32_t)(i * S);
                    float rs = 0.f;
                    // Pd = P keep (bf16, kept packed for the image rounds: prr <- its bits would cost registers, so it is
                    // recomputed per round from prr and the keep factor folded into acc's sign-free companion below)
                    h4v pdp[T][4];
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * g4 + e;
                                const int j = 32 * t + 8 * g4 + 4 * h2 + e;
                                const float keep = p.has_drop ? (drop_keep(flat_i + (uint32_t)j, dkn) ? dkn.scale : 0.f) : 1.f;
                                const float dpk = acc[t][r] * keep;
                                rs += dpk * prr[t][r];
                                acc[t][r] = dpk;
                                pdp[t][g4][e] = (__bf16)(prr[t][r] * keep);
                            }
                        }
                    rs += __shfl_xor(rs, 32, 64);
                    // dA = P (dP keep - rs): bias-table gradient, then bf16 as the A operand of dQ
#pragma unroll
                    for (int t = 0; t < T; ++t) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float da = prr[t][r] * (acc[t][r] - rs);
                            acc[t][r] = da;
                            dsum[t][r] += da;          // bias-table gradient: summed per (i, j) pair over this workgroup's sequences first
                        }
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                            for (int e = 0; e < 8; ++e) dAf[t][s2][e] = (__bf16)acc[t][8 * s2 + e];
                    }
                    // 2 T image rounds: key panel t of Pd, then of dA, written by every wave, read back transposed by wave t
#pragma unroll
                    for (int t = 0; t < T; ++t) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) *reinterpret_cast<h4v*>(img_w + 16 * g4) = pdp[t][g4];
                        A3_LDS_BARRIER();
                        if (wave == t) {
#pragma unroll
                            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                                for (int s2 = 0; s2 < 2; ++s2)
                                    PdTf[ti][s2] = a3_tr(img_r + (32 * ti + 16 * s2) * ILD, img_r + (32 * ti + 16 * s2 + 4) * ILD);
                        }
                        A3_LDS_BARRIER();
                    }
#pragma unroll
                    for (int t = 0; t < T; ++t) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            h4v da4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) da4[e] = dAf[t][g4 >> 1][4 * (g4 & 1) + e];
                            *reinterpret_cast<h4v*>(img_w + 16 * g4) = da4;
                        }
                        A3_LDS_BARRIER();
                        if (wave == t) {
#pragma unroll
                            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                                for (int s2 = 0; s2 < 2; ++s2)
                                    dATf[ti][s2] = a3_tr(img_r + (32 * ti + 16 * s2) * ILD, img_r + (32 * ti + 16 * s2 + 4) * ILD);
                        }
                        A3_LDS_BARRIER();
                    }
                }
            } else {
                // ---- token contractions: two 32-column tiles of dQ (u < u_c2), dV (u < u_c3) or dK
                const int ph = u < u_c2 ? 0 : (u < u_c3 ? 1 : 2);
                const int ct = 2 * (u - (ph == 0 ? u_c1 : (ph == 1 ? u_c2 : u_c3)));
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const char* Bs = slot + hf * HALF;
                    floatx16 o[1];
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[0][r] = 0.f;
                    if (ph == 0) {
#pragma unroll
                        for (int t = 0; t < T; ++t)
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                const attn_h8 fb = a3_tr(Bs + t * 2048 + s2 * 1024 + troff[0], Bs + t * 2048 + s2 * 1024 + troff[1]);
                                o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, dAf[t][s2], o[0], 0, 0, 0);
                            }
                        a3_store_tile(o[0], p.scale, r_dQ, a3_row_off(r0 + (uint32_t)i, (uint32_t)p.kbq, 0), keyq, i < S, (uint32_t)(p.tq0 + ((h * p.dk) >> 5) + ct + hf), h2);
                    } else if (ph == 1) {
#pragma unroll
                        for (int t = 0; t < T; ++t)
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                const attn_h8 fb = a3_tr(Bs + t * 2048 + s2 * 1024 + trnat[0], Bs + t * 2048 + s2 * 1024 + trnat[1]);
                                o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, PdTf[t][s2], o[0], 0, 0, 0);
                            }
                        a3_store_tile(o[0], 1.f, r_dV, a3_row_off(r0 + (uint32_t)i, (uint32_t)p.kbv, 0), keyq, i < S, (uint32_t)(p.tv0 + ((h * p.dv) >> 5) + ct + hf), h2);
                    } else {
#pragma unroll
                        for (int t = 0; t < T; ++t)
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                const attn_h8 fb = a3_tr(Bs + t * 2048 + s2 * 1024 + trnat[0], Bs + t * 2048 + s2 * 1024 + trnat[1]);
                                o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, dATf[t][s2], o[0], 0, 0, 0);
                            }
                        a3_store_tile(o[0], p.scale, r_dK, a3_row_off(r0 + (uint32_t)i, (uint32_t)p.kbk, 0), keyq, i < S, (uint32_t)(p.tk0 + ((h * p.dk) >> 5) + ct + hf), h2);
                    }
                }
            }
            if (++cslot == NB) cslot = 0;
        }
    }
    if (has_bias) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t ix = (idxp[t][r >> 1] >> (16 * (r & 1))) & 0xFFFFu;
                if (ix != 0xFFFFu) atomicAdd(&tw[ix], dsum[t][r]);
            }
    }
    }
    if (has_bias) {
        __syncthreads();
        for (int x = threadIdx.x; x < p.table_rows; x += 64 * (T + 1)) {
            float v = tacc[x];
#pragma unroll
            for (int w = 1; w < T; ++w) v += tacc[w * p.table_rows + x];
            if (p.table_partials) p.dtable[((size_t)chunk * p.table_rows + x) * p.H + h] = v;
            else atomicAdd(&p.dtable[(size_t)x * p.H + h], v);
        }
    }
}

// =====================================================================================================
// The same lane-=-query structure on the EXACT-f32 MFMA (fp32 mode; f32 Q, K, V, O in the token-major [M, H d] layout of the
// projection GEMMs): logits^T = K (Q scale)^T leaves lane = query, the softmax is lane-local, the dropped probabilities ARE the
// B operand of O^T = V^T Pd^T (v_mfma_f32_32x32x2_f32 contracts k = lane half: accumulator register r of the two lane halves holds
// the key pair (j, j + 4), and the A operand reads V[j][c] / V[j + 4][c] with plain ds_read_b32 - no transposed read is needed for
// 4-byte elements).  Staging: 32-feature chunks of 128-B rows, 8 rows x 128 B per LDS-DMA instruction, 16-B chunk index XOR
// (row >> 1) & 7 on the source side (conflict-free ds_read_b128, as the second generation), the producer-wave ring of the
// packed-operand kernels.  O leaves as 16-B stores (4 consecutive columns of the lane's row per accumulator register group).
// =====================================================================================================
struct A3UnitF { const float *b0, *b1; uint32_t ld0, ld1; };      // two [SP rows][32 floats] halves: base of (row 0, column 0), row pitch

template <int T, int NB, typename UnitOf>
__device__ __forceinline__ void a3f_producer(int n_begin, int n_end, int U, int S, UnitOf unit_of) {
    constexpr int SP = 32 * T, HALF = SP * 128, SLOT = 2 * HALF, P = 8 * T;
    static_assert(P * (NB - 2) <= 63, "vmcnt is a 6-bit counter");
    const int lane = threadIdx.x & 63;
    const int total = (n_end - n_begin) * U;
    int is_n = n_begin, is_u = 0, is_slot = 0, issued = 0;
    auto issue_next = [&]() {
        if (issued >= total) return;
        const A3UnitF un = unit_of(is_n, is_u);
        const uint32_t lb = (uint32_t)(is_slot * SLOT);
#pragma unroll
        for (int pc = 0; pc < 4 * T; ++pc) {
            const uint32_t r = (uint32_t)min(8 * pc + (lane >> 3), S - 1);
            const uint32_t sw = ((((uint32_t)lane & 7u) ^ ((r >> 1) & 7u)) << 4);
            const uint32_t lb0 = __builtin_amdgcn_readfirstlane(lb + pc * 1024), lb1 = __builtin_amdgcn_readfirstlane(lb + HALF + pc * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(r * un.ld0 * 4u + sw), "s"(un.b0), "s"(lb0) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(r * un.ld1 * 4u + sw), "s"(un.b1), "s"(lb1) : "memory");
        }
        ++issued;
        if (++is_u == U) { is_u = 0; ++is_n; }
        if (++is_slot == NB) is_slot = 0;
    };
#pragma unroll 1
    for (int k = 0; k < NB - 1; ++k) issue_next();
    int g = 0;
#pragma unroll 1
    for (int n = n_begin; n < n_end; ++n)
#pragma unroll 1
        for (int u = 0; u < U; ++u) {
            const int rem = min(NB - 2, total - 1 - g);
            if (rem >= NB - 2) A3_WAIT_VM(P * (NB - 2));
            else if (NB > 3 && rem == NB - 3) A3_WAIT_VM(P * (NB > 3 ? NB - 3 : 0));
            else A3_WAIT_VM(0);
            __builtin_amdgcn_s_barrier();
            issue_next();
            ++g;
        }
}

// (second launch-bound argument: waves per SIMD the register budget must allow - 3 workgroups per CU at T = 2, 6 and more at T = 1)
template <int T, int NB>
__global__ void __launch_bounds__(64 * (T + 1), T == 1 ? 4 : (T == 2 ? 3 : 2)) attn_fwd3f_kernel(const AttnParams p) {
    const DropKey dkn = drop_key_now(p.dkey);
    constexpr int SP = 32 * T, HALF = SP * 128, SLOT = 2 * HALF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const char* ring = reinterpret_cast<const char*>(sm);
    const int h = (int)(blockIdx.x % (unsigned)p.H), chunk = (int)(blockIdx.x / (unsigned)p.H), S = p.S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, h2 = lane >> 5;
    const int n_begin = chunk * p.n_per_wg, n_end = min(p.N, n_begin + p.n_per_wg);
    const int nq = p.dk >> 5, nvp = p.dv >> 6, U = nq + nvp;
    if (wave == T) {
        a3f_producer<T, NB>(n_begin, n_end, U, S, [&](int n, int u) -> A3UnitF {
            A3UnitF un;
            if (u < nq) {
                un.b0 = p.Q + (size_t)n * S * p.ldq + (size_t)h * p.dk + 32 * u; un.ld0 = (uint32_t)p.ldq;
                un.b1 = p.K + (size_t)n * S * p.ldk + (size_t)h * p.dk + 32 * u; un.ld1 = (uint32_t)p.ldk;
            } else {
                un.b0 = p.V + (size_t)n * S * p.ldv + (size_t)h * p.dv + 64 * (u - nq); un.ld0 = un.ld1 = (uint32_t)p.ldv;
                un.b1 = un.b0 + 32;
            }
            return un;
        });
        return;
    }
    const int i = 32 * wave + l31;
    const bool has_bias = p.index_ld > 0 && S > 1;
    const __amdgpu_buffer_rsrc_t r_index = __builtin_amdgcn_make_buffer_rsrc(const_cast<int64_t*>(p.index), 0,
        has_bias ? (S - 1) * p.index_ld * 8 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_table = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.table), 0,
        has_bias ? (p.table_rows > 0 ? p.table_rows * p.H * 4 : 0x7fffffff) : 0, 0x00020000);
    float biasr[T][16];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * t + 8 * (r >> 2) + 4 * h2 + (r & 3);
            const bool pair = i >= 1 && j >= 1 && i < S && j < S;
            const uint32_t ix = __builtin_amdgcn_raw_buffer_load_b32(r_index, pair ? (uint32_t)((i - 1) * p.index_ld + (j - 1)) * 8u : 0xFFFFFFFFu, 0, 0);
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_table, ((ix * (uint32_t)p.H + (uint32_t)h) * 4u) | (pair ? 0u : 0xFFFFFFFFu), 0, 0));
            biasr[t][r] = b + (j < S ? 0.f : -INFINITY);
        }
    // fragment offsets: this lane half's 16-float run of row r of a staged chunk; V[j][c] of a staged tile
    const int keyq = (i >> 1) & 7, keyk = (l31 >> 1) & 7;
    int offq[4], offk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        offq[q] = i * 128 + (((4 * h2 + q) ^ keyq) << 4);
        offk[q] = l31 * 128 + (((4 * h2 + q) ^ keyk) << 4);
    }
    int cslot = 0;
    floatx16 acc[T];
#pragma unroll 1
    for (int n = n_begin; n < n_end; ++n) {
#pragma unroll 1
        for (int u = 0; u < U; ++u) {
            A3_LDS_BARRIER();
            const char* slot = ring + cslot * SLOT;
            if (u < nq) {
                if (u == 0) {
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
                }
                float fq[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const a3_f4 v = *reinterpret_cast<const a3_f4*>(slot + offq[q]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fq[4 * q + e] = v[e] * p.scale;
                }
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    float fk[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const a3_f4 v = *reinterpret_cast<const a3_f4*>(slot + HALF + t * 4096 + offk[q]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) fk[4 * q + e] = v[e];
                    }
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fk[s], fq[s], acc[t], 0, 0, 0);
                }
                if (u == nq - 1) {
                    float m = -INFINITY;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float x = acc[t][r] + biasr[t][r];
                            acc[t][r] = x;
                            m = fmaxf(m, x);
                        }
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float e = expf(acc[t][r] - m);
                            acc[t][r] = e;
                            sum += e;
                        }
                    sum += __shfl_xor(sum, 32, 64);
                    float* pr_row = p.probs + (((size_t)n * p.H + h) * S + (size_t)min(i, S - 1)) * S;
                    const uint32_t flat_i = ((uint32_t)n * p.H + h) * (uint32_t)(S * S) + (uint32_t)(i * S);
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int j0 = 32 * t + 8 * g4 + 4 * h2;
                            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                            f4u pv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pv[e] = acc[t][4 * g4 + e] / sum;
                            if (i < S) {          // dense [S][S] rows (the backward kernels read that layout): 4-B aligned 16-B stores
                                if (j0 + 3 < S) *reinterpret_cast<f4u*>(pr_row + j0) = pv;
                                else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (j0 + e < S) pr_row[j0 + e] = pv[e];
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = pv[e];
                                if (p.has_drop) v = drop_keep(flat_i + (uint32_t)(j0 + e), dkn) ? v * dkn.scale : 0.f;
                                acc[t][4 * g4 + e] = v;
                            }
                        }
                }
            } else {
                // ---- O^T[two 32-column tiles][queries of this wave] = V^T Pd^T
                const int ct = 2 * (u - nq);
                float* o_row = p.O + ((size_t)n * S + (size_t)min(i, S - 1)) * p.ldo + (size_t)h * p.dv;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const char* Vs = slot + hf * HALF;
                    floatx16 o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int jl = (r & 3) + 8 * (r >> 2) + 4 * h2;                    // key row inside the 32-key tile
                            const float a = *reinterpret_cast<const float*>(Vs + t * 4096 + jl * 128 + ((((l31 >> 2)) ^ ((jl >> 1) & 7)) << 4) + (l31 & 3) * 4);
                            o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc[t][r], o, 0, 0, 0);
                        }
                    if (i < S) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            a3_f4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = o[4 * g4 + e];
                            *reinterpret_cast<a3_f4*>(o_row + 32 * (ct + hf) + 8 * g4 + 4 * h2) = v;
                        }
                    }
                }
            }
            if (++cslot == NB) cslot = 0;
        }
    }
}

template <typename Kern>
void set_lds(Kern k, size_t lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace

namespace lstc_attn {

// ring depth per instantiation: (NB - 1) x slot x resident workgroups ~ 100-130 KB of DMA in flight per CU (block comment above)
int attn3_fwd_launch(const AttnParams& p, int T, int chunks, hipStream_t st) {
#define LSTC_FWD3(TT, NBB)                                                                 \
    do {                                                                                   \
        static LstcDevOnce once3;                                                          \
        const int dev3_ = once3.begin();                                                   \
        if (dev3_ >= 0) { set_lds(attn_fwd3_kernel<TT, NBB>, 160 * 1024); once3.end(dev3_); } \
        hipLaunchKernelGGL((attn_fwd3_kernel<TT, NBB>), dim3((unsigned)chunks * (unsigned)p.H), 64 * (TT + 1), \
                           (size_t)NBB * 2 * (32 * TT) * 64, st, p);                       \
    } while (0)
    if (T == 1) LSTC_FWD3(1, 6); else if (T == 2) LSTC_FWD3(2, 9); else if (T == 3) LSTC_FWD3(3, 6); else return LSTC_E_RANGE;
#undef LSTC_FWD3
    return lstc_launch_status();
}

int attn3f_fwd_launch(const AttnParams& p, int T, int chunks, hipStream_t st) {
#define LSTC_FWD3F(TT, NBB)                                                                \
    do {                                                                                   \
        static LstcDevOnce once3;                                                          \
        const int dev3_ = once3.begin();                                                   \
        if (dev3_ >= 0) { set_lds(attn_fwd3f_kernel<TT, NBB>, 160 * 1024); once3.end(dev3_); } \
        hipLaunchKernelGGL((attn_fwd3f_kernel<TT, NBB>), dim3((unsigned)chunks * (unsigned)p.H), 64 * (TT + 1), \
                           (size_t)NBB * 2 * (32 * TT) * 128, st, p);                      \
    } while (0)
    if (T == 1) LSTC_FWD3F(1, 3); else if (T == 2) LSTC_FWD3F(2, 3); else if (T == 3) LSTC_FWD3F(3, 3); else return LSTC_E_RANGE;
#undef LSTC_FWD3F
    return lstc_launch_status();
}

int attn3_bwd_launch(const AttnParams& p, int T, int chunks, hipStream_t st) {
#define LSTC_BWD3(TT, NBB)                                                                 \
    do {                                                                                   \
        const size_t lds3 = (size_t)NBB * 2 * (32 * TT) * 64 + (size_t)(32 * TT) * 72 + (size_t)TT * p.table_rows * sizeof(float); \
        if (lds3 > 160 * 1024) return LSTC_E_RANGE;                                        \
        static LstcDevOnce once3;                                                          \
        const int dev3_ = once3.begin();                                                   \
        if (dev3_ >= 0) { set_lds(attn_bwd3_kernel<TT, NBB>, 160 * 1024); once3.end(dev3_); } \
        hipLaunchKernelGGL((attn_bwd3_kernel<TT, NBB>), dim3((unsigned)chunks * (unsigned)p.H), 64 * (TT + 1), lds3, st, p); \
    } while (0)
    if (T == 1) LSTC_BWD3(1, 9); else if (T == 2) LSTC_BWD3(2, 8); else if (T == 3) LSTC_BWD3(3, 5); else return LSTC_E_RANGE;
#undef LSTC_BWD3
    return lstc_launch_status();
}

}  // namespace lstc_attn
