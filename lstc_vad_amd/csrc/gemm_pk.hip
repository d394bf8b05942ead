// f32-accurate GEMM on the 16-bit matrix cores over PACKED, pre-split operands (LstcGemmDesc.dtype = LSTC_F32X3).
//
// Arithmetic.  An operand tensor is scaled by a power of two s (exact) so that its largest magnitude lies in [2^14, 2^15),
// then every element is written x*s = h + l + d with h = f16(x*s), l = f16(x*s - h) (round-to-nearest-even): two 11-bit
// significands, |d| <= 2^-24 |x*s| - the representation error of an f32 itself - for every element within 2^-17 of the
// tensor's maximum (smaller ones keep an ABSOLUTE error below 2^-39 of the maximum).  A dot product is the sum of the
// three plane products {hh, hl, lh}; each f16 x f16 product is exact in f32, v_mfma_f32_32x32x16_f16 accumulates in f32
// and the epilogue multiplies by 1/(s_a s_b), again exact.  Measured against f64 (tools/x3_probe.hip, K = 2048,
// activations x weights): rms error 0.82e-7 of the result vs 0.92e-7 for a sequential f32 fma chain - f32 accuracy.
// The 16-bit MFMA runs at 16x the rate of v_mfma_f32_32x32x2_f32, so three per k16 block cost 96 cycles against 512.
// (A first version used three bf16 planes and six products - no scaling, 1.6x slower; tools/x3_probe.hip keeps it.)
//
// Data movement.  Splitting rewrites an operand anyway, so the same pass PACKS it (lstc_pack3): the matrix is cut into
// 128-row x 32-k tiles and each plane of a tile is stored as the exact 8-KB LDS image the kernel reads - unpadded rows of
// four 16-B chunks, chunk index XOR-ed with (row >> 2) & 3, which puts the 16 lanes of every ds_read_b128 lane group on
// 16 distinct 16-B slots (MI355X_MICROARCH, LDS table).  Transposition happens in the pack pass too, so forward (X W^T)
// and input-gradient (dY W) products share ONE NT kernel; weight gradients (dY^T X) reuse those very packs through
// ds_read_b64_tr_b16 (TR form).  The GEMM streams tiles with global_load_lds_dwordx4 (1 KB contiguous per
// wave-instruction, no staging registers, no ds_write) into a 3-stage LDS ring: tile t+3 is requested during the second
// half of tile t, two K tiles ahead of its use.
//
// Schedule per K tile and wave (one wave per SIMD, 4 waves, 128x128 tile, 64x64 per wave): phase 1 = 12 MFMAs on k-step 0
// fragments while the k-step 1 fragments are read; s_waitcnt vmcnt(8) + raw s_barrier (tile t+1 published, stage t free);
// phase 2 = 12 MFMAs on k-step 1 while tile t+3 is requested and the k-step 0 fragments of tile t+1 are read.
#include "lstc_common.h"

namespace {

typedef _Float16 pk_t;
typedef _Float16 pkx8 __attribute__((ext_vector_type(8)));

constexpr int NT = 256;
constexpr int PK_IMG = 4096;            // f16 elements of one plane image (128 rows x 32 k)
constexpr int PK_TILE = 2 * PK_IMG;     // one packed tile: planes h, l
constexpr int PK_STAGE = 2 * PK_TILE;   // A tile then B tile
constexpr int PK_NSTAGE = 3;
constexpr int PK_TRAILER = 256;         // bytes after the tiles: [0] absmax bits (u32), [1] 1/scale (f32)

__device__ __forceinline__ void split2(float v, float scale, pk_t& h, pk_t& l) {
    const float xs = v * scale;
    h = (pk_t)xs;
    l = (pk_t)(xs - (float)h);
}

// scale = 2^(14 - floor(log2(absmax))): the tensor's largest magnitude lands in [2^14, 2^15) (f16 max = 65504)
__device__ __forceinline__ float scale_from_absmax_bits(uint32_t bits) {
    int e = (int)((bits >> 23) & 0xff) - 127;             // floor(log2(absmax)) for normal numbers
    if (bits == 0u || e > 127) return 1.f;                // all-zero tensor (or inf / nan): leave as is
    int se = 14 - e;
    se = se > 126 ? 126 : (se < -126 ? -126 : se);
    return __uint_as_float((uint32_t)(se + 127) << 23);
}

__global__ void __launch_bounds__(NT) absmax_kernel(const float* __restrict__ x, long long rows, long long cols, long long ld,
                                                    uint32_t* __restrict__ out) {
    float m = 0.f;
    const long long n = rows * cols;
    if (ld == cols && (n & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {      // contiguous: float4 stream
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < (n >> 2); i += (long long)gridDim.x * NT) {
            const float4 v = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (long long r = blockIdx.x; r < rows; r += gridDim.x)
            for (long long c = threadIdx.x; c < cols; c += NT) m = fmaxf(m, fabsf(x[r * ld + c]));
    }
    // one atomic per workgroup: thousands of atomics on ONE address serialise (~10 ns each)
    __shared__ float red[NT / 64];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) mm = fmaxf(mm, red[w]);
        atomicMax(out, __float_as_uint(mm));                               // non-negative floats order like their bits
    }
}

// ---- pack, K-contiguous source [rows, K] (ld): one workgroup per row block and PK_KPB consecutive k blocks; per tile a
// thread converts two 32-B source chunks.  All 8 * 2 loads of a thread are issued before the first store (memory-level
// parallelism: one tile per workgroup left the kernel latency-bound at 4.3 TB/s).
constexpr int PK_KPB = 4;
__global__ void __launch_bounds__(NT) pack3_kc_kernel(const float* __restrict__ x, int rows, int K, long long ld,
                                                      pk_t* __restrict__ out, int KB, uint32_t* __restrict__ trailer) {
    const int kgroups = (KB + PK_KPB - 1) / PK_KPB;
    const int kb0 = (blockIdx.x % kgroups) * PK_KPB, rb = blockIdx.x / kgroups;
    const float scale = scale_from_absmax_bits(trailer[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(trailer)[1] = 1.f / scale;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    float4 va[PK_KPB][2], vb[PK_KPB][2];
#pragma unroll
    for (int tI = 0; tI < PK_KPB; ++tI)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s = threadIdx.x + q * NT;      // chunk id inside the tile: row r = s / 4, chunk c = s % 4
            const int row = rb * 128 + (s >> 2), k0 = (kb0 + tI) * 32 + (s & 3) * 8;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a;
            if (row < rows && kb0 + tI < KB) {
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
    for (int tI = 0; tI < PK_KPB; ++tI) {
        if (kb0 + tI >= KB) break;
        pkx8* o = reinterpret_cast<pkx8*>(out) + ((size_t)rb * KB + kb0 + tI) * 2 * 512;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s = threadIdx.x + q * NT, r = s >> 2, c = s & 3;
            const float v[8] = {va[tI][q].x, va[tI][q].y, va[tI][q].z, va[tI][q].w, vb[tI][q].x, vb[tI][q].y, vb[tI][q].z, vb[tI][q].w};
            pkx8 hh, ll;
#pragma unroll
            for (int j = 0; j < 8; ++j) { pk_t h, l; split2(v[j], scale, h, l); hh[j] = h; ll[j] = l; }
            const int slot = r * 4 + (c ^ ((r >> 2) & 3));
            o[slot] = hh; o[512 + slot] = ll;
        }
    }
}

// ---- pack, k-major source [K, rows] (ld): the packed operand's row index runs along the source's contiguous dimension.
// Thread -> one packed row (feature) and 16 of the tile's 32 k (source rows): coalesced reads along the features.
__global__ void __launch_bounds__(NT) pack3_km_kernel(const float* __restrict__ x, int rows, int K, long long ld,
                                                      pk_t* __restrict__ out, int KB, uint32_t* __restrict__ trailer) {
    const int kb = blockIdx.x % KB, rb = blockIdx.x / KB;
    const float scale = scale_from_absmax_bits(trailer[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(trailer)[1] = 1.f / scale;
    pkx8* o = reinterpret_cast<pkx8*>(out) + (size_t)blockIdx.x * 2 * 512;
    const int r = threadIdx.x & 127, half = threadIdx.x >> 7;
    const int row = rb * 128 + r;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int c = half * 2 + cc, k0 = kb * 32 + c * 8;
        pkx8 hh, ll;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (row < rows && k0 + j < K) ? x[(size_t)(k0 + j) * ld + row] : 0.f;
            pk_t h, l; split2(v, scale, h, l); hh[j] = h; ll[j] = l;
        }
        const int slot = r * 4 + (c ^ ((r >> 2) & 3));
        o[slot] = hh; o[512 + slot] = ll;
    }
}

struct PkParams {
    const pk_t* A;
    const pk_t* B;
    float* C;
    const float* bias;
    const float* res;
    const float* relu_src;
    int M, N, ldc, ldr, ld_relu, flags;
    float alpha;
    DropKey dk;
    int tilesN, KB, ktiles_per_split;
    const float* inv_a;  // 1/scale of the A and B packs (their trailers)
    const float* inv_b;
    long long split_stride;   // elements between the outputs of consecutive K splits (0: add into one C with atomics)
    int fbA, fbB;        // TR mode: 32-feature blocks per token row-block of the A / B packs (= ceil(M/32), ceil(N/32))
    int epi_f4;          // 0: scalar epilogue; 1 / 2: float4 epilogue (lstc_common.h, epilogue_f4) without / with one per-element operand
};

#ifndef PK_GROUP_M
#define PK_GROUP_M 8          /* grouped tile order inside an XCD's run (as csrc/gemm_f32.hip); 0 = row-major, the order of rounds 1-3.
                                 Same-box A/B, round 4: 100352 x 4096 x 2048 4.44 -> 4.13 ms, x 2048 x 2048 2.16 -> 2.14, weight gradient
                                 2.22 -> 2.15, f32x3 LTN step 117.1 / 117.2 -> 115.6 / 115.7 ms; bit-identical products.  Round 5: 2 / 4 / 16 against 8 on the
                                 step's four forward shapes: within 2 % of each other (16 is 8 % slower at K = 4096) although the launch
                                 moves 7.3 GB for 1.7 GB of operands (profiles/r05_gemm_pk_pmc.txt): the re-fetches are Infinity-Cache hits */
#endif
// linear tile index (after the XCD remap) -> (M tile, N tile): row-major, or groups of PK_GROUP_M consecutive M tiles walked M
// fastest - the 64 tiles resident on an XCD then cover 8 x 8 tiles instead of 4 x 16 and share twice as much per K step
__device__ __forceinline__ void pk_tile_of(int pid, int tilesM, int tilesN, int& mb, int& nb) {
#if PK_GROUP_M
    const int per_group = PK_GROUP_M * tilesN;
    const int gid = pid / per_group, first_m = gid * PK_GROUP_M;
    const int gsz = min(tilesM - first_m, PK_GROUP_M);
    const int loc = pid - gid * per_group;
    mb = first_m + loc % gsz;
    nb = loc / gsz;
#else
    mb = pid / tilesN;
    nb = pid % tilesN;
#endif
}


// plane pairs by decreasing magnitude: q = 0: hh, 1: hl, 2: lh
__device__ __forceinline__ constexpr int pa(int q) { return q == 2 ? 1 : 0; }
__device__ __forceinline__ constexpr int pb(int q) { return q == 1 ? 1 : 0; }

template <int SMODE, int CUR>
struct StepTag { static constexpr int smode = SMODE, cur = CUR; };

// TR = false: operands packed as [M, K] / [N, K] (K along the 32-wide tile dimension), fragments by ds_read_b128.
// TR = true (weight gradients, C = dY^T X): operands are the packs of the SOURCE matrices [K = tokens, M] / [K, N] as the
// forward / input-gradient products already made them; a K step is 32 tokens = a contiguous 2-KB slice of each plane image,
// and the fragments (8 consecutive tokens of one feature) come out of ds_read_b64_tr_b16, the hardware transpose read.
template <bool TR>
__global__ void __launch_bounds__(NT, 1) gemm_pk_kernel(const PkParams p) {
    const DropKey dkn = drop_key_now(p.dk);
    extern __shared__ __attribute__((aligned(16))) pk_t smem_pk[];
    pk_t* const smem = smem_pk;
    int pid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = pid & 7, idx = pid >> 3, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int mb, nb;
    pk_tile_of(pid, (int)gridDim.x / p.tilesN, p.tilesN, mb, nb);
    const int kt0 = blockIdx.y * p.ktiles_per_split;
    const int nkt = min(p.KB, kt0 + p.ktiles_per_split) - kt0;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // a stage is 32 pieces of 1 KB: pieces 0..15 = the A tile, 16..31 = the B tile; wave w moves pieces 8w .. 8w+7
    // (waves 0,1: A; waves 2,3: B).  gbase is wave-uniform (SGPRs); the lane adds 16 B * lane.
    //   TR = false: the 16 KB of a packed tile are contiguous; piece j = 4 * (j / 4) + (j % 4), the low part rides in
    //     the instruction's immediate offset (it applies to the global and the LDS side alike).
    //   TR = true: K step kt = tokens 32 kt .. 32 kt + 31 = slice (kt % 4) of token block kt / 4; for each of the wave's
    //     two 32-feature blocks and two planes one 2-KB slice = 2 pieces (immediate offset 0 / 1024).
    const int opb = wave < 2 ? mb : nb, fb = wave < 2 ? p.fbA : p.fbB;
    const pk_t* gbase = TR ? (wave < 2 ? p.A : p.B) + ((size_t)opb * 4 + (wave & 1) * 2) * PK_TILE
                             : (wave < 2 ? p.A + ((size_t)mb * p.KB + kt0) * PK_TILE : p.B + ((size_t)nb * p.KB + kt0) * PK_TILE) +
                                   (size_t)(wave & 1) * 8 * 512;
    const int ldst = wave * 8 * 512;
    auto koff = [&](int kt) -> size_t {      // element offset of K step kt (relative to kt0 for TR = false)
        if (TR) { const int k = kt0 + kt; return ((size_t)(k >> 2) * fb) * PK_TILE + (size_t)(k & 3) * 1024; }
        return (size_t)kt * PK_TILE;
    };
    // TR = true issues the DMA through inline asm: with the builtin the compiler knows these instructions write LDS, cannot
    // tell the ds_read_b64_tr_b16 fragment reads apart from their destinations and drains vmcnt(0) before every read
    // (2.7x slower).  All vmcnt waits of this kernel are explicit anyway (s_waitcnt before the barriers).
#define DMA_ONE(j, kt, stage)                                                                                          \
    do {                                                                                                               \
        if (TR) {                                                                                                      \
            const pk_t* g_ = gbase + koff(kt) + ((j) >> 2) * PK_TILE + (((j) & 3) >> 1) * PK_IMG + lane * 8;            \
            const uint32_t l_ = (uint32_t)(((stage) * PK_STAGE + ldst + ((j) >> 1) * 1024) * 2);                        \
            asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off offset:%2"                               \
                         :: "v"(g_), "s"(l_), "n"(((j) & 1) * 1024) : "memory");                                       \
        } else                                                                                                         \
            __builtin_amdgcn_global_load_lds((gptr_t)(gbase + koff(kt) + ((j) >> 2) * 2048 + lane * 8),                 \
                                             (lptr_t)(smem + (stage) * PK_STAGE + ldst + ((j) >> 2) * 2048), 16, ((j) & 3) * 1024, 0); \
    } while (0)
#define DMA_TILE(kt, stage)                                                                                            \
    do {                                                                                                               \
        DMA_ONE(0, kt, stage); DMA_ONE(1, kt, stage); DMA_ONE(2, kt, stage); DMA_ONE(3, kt, stage);                    \
        DMA_ONE(4, kt, stage); DMA_ONE(5, kt, stage); DMA_ONE(6, kt, stage); DMA_ONE(7, kt, stage);                    \
    } while (0)
    // TR = false: fragment of k-step ks = row tile0 + l31, logical 16-B chunk 2h + ks (any K permutation shared by A, B).
    // TR = true: stage layout per operand [feature block 0..3][plane][32 tokens x 64 B]; lane (h, g = 16-lane group & 1,
    //   q, pp) addresses token 16 ks + 8 h + q (+4), features 16 g + 4 pp .. + 3 and receives feature lane & 31.
    const int rowa0 = wm * 64 + l31, rowb0 = wn * 64 + l31;
    const int trq = (lane >> 2) & 3, trchunk = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), trsub = (lane & 1) * 4;
    auto rd = [&](const pk_t* img, int row, int ks) -> pkx8 {
        return *reinterpret_cast<const pkx8*>(img + (row * 4 + ((2 * h + ks) ^ ((row >> 2) & 3))) * 8);
    };
    auto rd_tr = [&](const pk_t* img, int ks) -> pkx8 {
        typedef short short4v __attribute__((ext_vector_type(4)));
        typedef short short8v __attribute__((ext_vector_type(8)));
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const int t0 = 16 * ks + 8 * h + trq, t1 = t0 + 4;
        const pk_t* a0 = img + t0 * 32 + ((trchunk ^ ((t0 >> 2) & 3)) * 8) + trsub;
        const pk_t* a1 = img + t1 * 32 + ((trchunk ^ ((t1 >> 2) & 3)) * 8) + trsub;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a1));
        short8v f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return __builtin_bit_cast(pkx8, f);
    };
    pkx8 f0a[2][2], f0b[2][2], f1a[2][2], f1b[2][2];
    // read order = order of first use by the plane-pair rounds (lh, hl, hh): A.l, B.h, A.h, B.l
    auto frag_one = [&](int e, const pk_t* s, int ks, pkx8 (&fa)[2][2], pkx8 (&fb)[2][2]) {
        const int g = e >> 1, i = e & 1;
        const int pl = (g == 0 || g == 3) ? 1 : 0;
        if (TR) {
            if ((g & 1) == 0) fa[pl][i] = rd_tr(s + ((2 * wm + i) * 2 + pl) * 1024, ks);
            else fb[pl][i] = rd_tr(s + PK_TILE + ((2 * wn + i) * 2 + pl) * 1024, ks);
        } else {
            if ((g & 1) == 0) fa[pl][i] = rd(s + pl * PK_IMG, rowa0 + i * 32, ks);
            else fb[pl][i] = rd(s + (2 + pl) * PK_IMG, rowb0 + i * 32, ks);
        }
    };
    // ---- prologue: tiles 0, 1, 2 -> stages 0, 1, 2 (K index clamped: the in-order vmcnt bookkeeping is then the same on
    // every path into the loop; sched_barriers keep the issue order)
    DMA_TILE(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    DMA_TILE(min(1, nkt - 1), 1);
    __builtin_amdgcn_sched_barrier(0);
    DMA_TILE(min(2, nkt - 1), 2);
    __builtin_amdgcn_s_waitcnt(0x4F70);              // vmcnt(16): tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int e = 0; e < 8; ++e) frag_one(e, smem, 0, f0a, f0b);

    auto step = [&](int it, auto tag) {
        // SMODE 1: tiles t+1..t+3 exist; 4: t+1, t+2 exist (nothing more to request); 2: only t+1; 3: last tile
        constexpr int SMODE = decltype(tag)::smode;
        constexpr int CUR = decltype(tag)::cur;              // stage of tile t (0..2)
        constexpr int NXT = (CUR + 1) % 3;
        constexpr bool HAS1 = SMODE != 3, HAS3 = SMODE == 1;
        const pk_t* s_cur = smem + CUR * PK_STAGE;
        const pk_t* s_nxt = smem + NXT * PK_STAGE;
#define PK_MMA(FA, FB, q)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa(q)][i], FB[pb(q)][j], acc[i][j], 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0)
    // three rounds of 4 MFMAs per phase; the 8 fragment reads / 8 DMA pieces of a phase are dealt 3, 3, 2
#define PK_R1(r)                                                                                            \
    frag_one(3 * (r), s_cur, 1, f1a, f1b); frag_one(3 * (r) + 1, s_cur, 1, f1a, f1b);                        \
    if ((r) < 2) frag_one(3 * (r) + 2, s_cur, 1, f1a, f1b);                                                  \
    PK_MMA(f0a, f0b, 2 - (r))
#define PK_R2(r)                                                                                            \
    if (HAS3) { DMA_ONE(3 * (r), it + 3, CUR); DMA_ONE(3 * (r) + 1, it + 3, CUR); }                          \
    if (HAS3 && (r) < 2) { DMA_ONE(((r) < 2 ? 3 * (r) + 2 : 0), it + 3, CUR); }                              \
    if (HAS1) { frag_one(3 * (r), s_nxt, 0, f0a, f0b); frag_one(3 * (r) + 1, s_nxt, 0, f0a, f0b); }          \
    if (HAS1 && (r) < 2) frag_one(3 * (r) + 2, s_nxt, 0, f0a, f0b);                                          \
    PK_MMA(f1a, f1b, 2 - (r))
        PK_R1(0); PK_R1(1); PK_R1(2);
        // tile t+1 must have landed (requested two K tiles ago); tile t+2's 8 requests may stay in flight.
        // Raw s_barrier: __syncthreads() adds a fence that drains EVERY LDS-DMA in flight (vmcnt(0)).
        if (SMODE == 1 || SMODE == 4) __builtin_amdgcn_s_waitcnt(0x0078);     // vmcnt(8) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        PK_R2(0); PK_R2(1); PK_R2(2);
#undef PK_MMA
#undef PK_R1
#undef PK_R2
    };
    int it = 0;
    for (; it + 5 < nkt; it += 3) {
        step(it, StepTag<1, 0>{});
        step(it + 1, StepTag<1, 1>{});
        step(it + 2, StepTag<1, 2>{});
    }
    for (; it < nkt; it += 3) {      // tail (it % 3 == 0): 1..5 tiles left
        const int rem = nkt - it;
        if (rem >= 4) step(it, StepTag<1, 0>{}); else if (rem == 3) step(it, StepTag<4, 0>{}); else if (rem == 2) step(it, StepTag<2, 0>{}); else step(it, StepTag<3, 0>{});
        if (rem >= 5) step(it + 1, StepTag<1, 1>{}); else if (rem == 4) step(it + 1, StepTag<4, 1>{}); else if (rem == 3) step(it + 1, StepTag<2, 1>{}); else if (rem == 2) step(it + 1, StepTag<3, 1>{});
        if (rem >= 6) step(it + 2, StepTag<1, 2>{}); else if (rem == 5) step(it + 2, StepTag<4, 2>{}); else if (rem == 4) step(it + 2, StepTag<2, 2>{}); else if (rem == 3) step(it + 2, StepTag<3, 2>{});
    }
#undef DMA_ONE
#undef DMA_TILE

    // ---- epilogue (semantics of gemm_f32.hip)
    const int flags = p.flags;
    const bool atomic = gridDim.y > 1 && p.split_stride == 0;      // split_stride != 0: split z owns C + z * split_stride
    float* const Cz = p.C + (size_t)blockIdx.y * p.split_stride;
    const float alpha = p.alpha * p.inv_a[0] * p.inv_b[0];
    if (p.epi_f4 && !atomic) {                 // float4 form (host checked alignment and the operand count)
        EpiArgs ea = make_epi_args(Cz, p.bias, p.res, p.relu_src, p.M, p.N, p.ldc, p.ldr, p.ld_relu, flags, 0, alpha, dkn);
        if (p.epi_f4 == 2) epilogue_f4<2, 2, true>(ea, acc, mb * 128 + wm * 64, nb * 128 + wn * 64, lane);
        else epilogue_f4<2, 2, false>(ea, acc, mb * 128 + wm * 64, nb * 128 + wn * 64, lane);
        return;
    }       // undo the operands' power-of-two scales (exact)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = nb * 128 + wn * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = (flags & LSTC_EPI_BIAS) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rbase = mb * 128 + wm * 64 + i * 32 + 4 * h;
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
                    v = drop_keep(idx, dkn) ? v * dkn.scale : 0.f;
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
// Two-stage, two-workgroups-per-CU form of the 128x128 kernel (64 KB of LDS and <= 256 registers per workgroup): the
// prologue (tile requests before the first MFMA) and the epilogue of one workgroup overlap the K loop of the other,
// which matters for the K = 2048 products (64 K steps per workgroup).

template <bool TR>
__global__ void __launch_bounds__(NT, 2) gemm_pk2s_kernel(const PkParams p) {
    const DropKey dkn = drop_key_now(p.dk);
    extern __shared__ __attribute__((aligned(16))) pk_t smem_pk[];
    pk_t* const smem = smem_pk;
    int pid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = pid & 7, idx = pid >> 3, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int mb, nb;
    pk_tile_of(pid, (int)gridDim.x / p.tilesN, p.tilesN, mb, nb);
    const int kt0 = blockIdx.y * p.ktiles_per_split;
    const int nkt = min(p.KB, kt0 + p.ktiles_per_split) - kt0;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // pieces as in gemm_pk_kernel: a stage = [A tile 16 KB][B tile 16 KB]; wave w moves 8 pieces of its operand's tile
    const int opb = wave < 2 ? mb : nb, fb = wave < 2 ? p.fbA : p.fbB;
    const pk_t* gbase = TR ? (wave < 2 ? p.A : p.B) + ((size_t)opb * 4 + (wave & 1) * 2) * PK_TILE
                           : (wave < 2 ? p.A + ((size_t)mb * p.KB + kt0) * PK_TILE : p.B + ((size_t)nb * p.KB + kt0) * PK_TILE) +
                                 (size_t)(wave & 1) * 8 * 512;
    const int ldst = wave * 8 * 512;
    auto koff = [&](int kt) -> size_t {
        if (TR) { const int k = kt0 + kt; return ((size_t)(k >> 2) * fb) * PK_TILE + (size_t)(k & 3) * 1024; }
        return (size_t)kt * PK_TILE;
    };
    // piece j (0..7).  NT: contiguous, j = 4 (j / 4) + (j % 4) with the low part as immediate offset.  TR: feature block
    // j / 4, plane (j % 4) / 2, half j % 2.  SGPR base + 32-bit lane offset (saddr form), inline asm (see gemm_pk_kernel).
    const uint32_t lane_off = (uint32_t)lane * 16u;
#define S2_DMA(j, kt, stage)                                                                                           \
    do {                                                                                                               \
        const pk_t* g_ = TR ? gbase + koff(kt) + ((j) >> 2) * PK_TILE + (((j) & 3) >> 1) * PK_IMG                       \
                            : gbase + koff(kt) + ((j) >> 2) * 2048;                                                     \
        const uint32_t l_ = (uint32_t)(((stage) * PK_STAGE + ldst + (TR ? ((j) >> 1) * 1024 : ((j) >> 2) * 2048)) * 2); \
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"                                    \
                     :: "v"(lane_off), "s"(g_), "s"(l_), "n"(TR ? ((j) & 1) * 1024 : ((j) & 3) * 1024) : "memory");    \
    } while (0)
#define S2_DMA_TILE(kt, stage)                                                                                         \
    do {                                                                                                               \
        S2_DMA(0, kt, stage); S2_DMA(1, kt, stage); S2_DMA(2, kt, stage); S2_DMA(3, kt, stage);                        \
        S2_DMA(4, kt, stage); S2_DMA(5, kt, stage); S2_DMA(6, kt, stage); S2_DMA(7, kt, stage);                        \
    } while (0)
    const int trq = (lane >> 2) & 3, trchunk = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), trsub = (lane & 1) * 4;
    auto rd = [&](const pk_t* img, int row, int ks) -> pkx8 {
        return *reinterpret_cast<const pkx8*>(img + (row * 4 + ((2 * h + ks) ^ ((row >> 2) & 3))) * 8);
    };
    auto rd_tr = [&](const pk_t* img, int ks) -> pkx8 {
        typedef short short4v __attribute__((ext_vector_type(4)));
        typedef short short8v __attribute__((ext_vector_type(8)));
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const int t0 = 16 * ks + 8 * h + trq, t1 = t0 + 4;
        const pk_t* a0 = img + t0 * 32 + ((trchunk ^ ((t0 >> 2) & 3)) * 8) + trsub;
        const pk_t* a1 = img + t1 * 32 + ((trchunk ^ ((t1 >> 2) & 3)) * 8) + trsub;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a1));
        short8v f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return __builtin_bit_cast(pkx8, f);
    };
    const int rowa0 = wm * 64 + l31, rowb0 = wn * 64 + l31;
    pkx8 f0a[2][2], f0b[2][2], f1a[2][2], f1b[2][2];
    // fragment e (0..7) in order of first use by the rounds (lh, hl, hh): A.l, B.h, A.h, B.l (two each)
    auto frag_one = [&](int e, const pk_t* s, int ks, pkx8 (&fa)[2][2], pkx8 (&fb)[2][2]) {
        const int g = e >> 1, i = e & 1;
        const int pl = (g == 0 || g == 3) ? 1 : 0;
        if (TR) {
            if ((g & 1) == 0) fa[pl][i] = rd_tr(s + ((2 * wm + i) * 2 + pl) * 1024, ks);
            else fb[pl][i] = rd_tr(s + PK_TILE + ((2 * wn + i) * 2 + pl) * 1024, ks);
        } else {
            if ((g & 1) == 0) fa[pl][i] = rd(s + pl * PK_IMG, rowa0 + i * 32, ks);
            else fb[pl][i] = rd(s + (2 + pl) * PK_IMG, rowb0 + i * 32, ks);
        }
    };
    // ---- prologue: tiles 0, 1 -> stages 0, 1 (K index clamped so that every path into the loop has the same vmcnt state)
    S2_DMA_TILE(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    S2_DMA_TILE(min(1, nkt - 1), 1);
    __builtin_amdgcn_s_waitcnt(0x0F78);              // vmcnt(8): tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int e = 0; e < 8; ++e) frag_one(e, smem, 0, f0a, f0b);

    auto step = [&](int it, auto tag) {
        // SMODE 1: tiles t+1, t+2 exist; 2: only t+1 (nothing more to request); 3: last tile
        constexpr int SMODE = decltype(tag)::smode;
        constexpr int CUR = decltype(tag)::cur;              // stage of tile t (0..1)
        constexpr bool HAS1 = SMODE != 3, HAS2 = SMODE == 1;
        const pk_t* s_cur = smem + CUR * PK_STAGE;
        const pk_t* s_nxt = smem + (CUR ^ 1) * PK_STAGE;
        // a round = product q on the four accumulators: 4 independent MFMAs; 3 rounds per phase
#define S2_MMA(FA, FB, q)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa(q)][i], FB[pb(q)][j], acc[i][j], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0)
#define S2_R1(r)                                                                                                \
    frag_one(3 * (r), s_cur, 1, f1a, f1b); frag_one(3 * (r) + 1, s_cur, 1, f1a, f1b);                            \
    if ((r) < 2) frag_one(3 * (r) + 2, s_cur, 1, f1a, f1b);                                                      \
    S2_MMA(f0a, f0b, 2 - (r))
        S2_R1(0); S2_R1(1); S2_R1(2);
        // tile t+1 (requested one and a half phases ago) must have landed: nothing newer is in flight.  The second
        // workgroup of the CU covers what this wait exposes.
        __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
#define S2_R2(r)                                                                                                \
    if (HAS2) { S2_DMA(3 * (r), it + 2, CUR); S2_DMA(3 * (r) + 1, it + 2, CUR); }                                 \
    if (HAS2 && (r) < 2) { S2_DMA(((r) < 2 ? 3 * (r) + 2 : 0), it + 2, CUR); }                                   \
    if (HAS1) { frag_one(3 * (r), s_nxt, 0, f0a, f0b); frag_one(3 * (r) + 1, s_nxt, 0, f0a, f0b); }              \
    if (HAS1 && (r) < 2) frag_one(3 * (r) + 2, s_nxt, 0, f0a, f0b);                                              \
    S2_MMA(f1a, f1b, 2 - (r))
        S2_R2(0); S2_R2(1); S2_R2(2);
#undef S2_MMA
#undef S2_R1
#undef S2_R2
    };
    int it = 0;
    for (; it + 3 < nkt; it += 2) {
        step(it, StepTag<1, 0>{});
        step(it + 1, StepTag<1, 1>{});
    }
    {   // tail (it even): 1..3 tiles left
        const int rem = nkt - it;
        if (rem == 3) { step(it, StepTag<1, 0>{}); step(it + 1, StepTag<2, 1>{}); step(it + 2, StepTag<3, 0>{}); }
        else if (rem == 2) { step(it, StepTag<2, 0>{}); step(it + 1, StepTag<3, 1>{}); }
        else if (rem == 1) { step(it, StepTag<3, 0>{}); }
    }
#undef S2_DMA
#undef S2_DMA_TILE

    const int flags = p.flags;
    const bool atomic = gridDim.y > 1 && p.split_stride == 0;      // split_stride != 0: split z owns C + z * split_stride
    float* const Cz = p.C + (size_t)blockIdx.y * p.split_stride;
    const float alpha = p.alpha * p.inv_a[0] * p.inv_b[0];
    if (p.epi_f4 && !atomic) {                 // float4 form (host checked alignment and the operand count)
        EpiArgs ea = make_epi_args(Cz, p.bias, p.res, p.relu_src, p.M, p.N, p.ldc, p.ldr, p.ld_relu, flags, 0, alpha, dkn);
        if (p.epi_f4 == 2) epilogue_f4<2, 2, true>(ea, acc, mb * 128 + wm * 64, nb * 128 + wn * 64, lane);
        else epilogue_f4<2, 2, false>(ea, acc, mb * 128 + wm * 64, nb * 128 + wn * 64, lane);
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = nb * 128 + wn * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = (flags & LSTC_EPI_BIAS) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rbase = mb * 128 + wm * 64 + i * 32 + 4 * h;
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
                    v = drop_keep(idx, dkn) ? v * dkn.scale : 0.f;
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
// 256x128 tile, 4 waves x (128x64), three LDS stages of 48 KB, one workgroup per CU.  The 128x128 kernels move 4 B per 64
// MACs out of L2 and sit at ~10.5 TB/s of L2->LDS traffic; this form moves 3/4 of that and reads 12 fragments per 24 MFMAs
// instead of 8 per 12.  A wave's 128 rows of A are one packed row block.  Stage = [A block 0][A block 1][B block], 48
// pieces of 1 KB, piece P = 12 wave + j.  A round = one plane product on one 32-row band = 2 MFMAs + one fragment read
// (+ one DMA piece in phase 2).  TR form: stage = [A: 8 feature blocks x 2 planes x 2 KB][B: 4 x 2 x 2 KB].
constexpr int WD_STAGE = 3 * PK_TILE;       // 48 KB

template <bool TR>
__global__ void __launch_bounds__(NT, 1) gemm_pkw_kernel(const PkParams p) {
    const DropKey dkn = drop_key_now(p.dk);
    extern __shared__ __attribute__((aligned(16))) pk_t smem_pk[];
    pk_t* const smem = smem_pk;
    int pid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = pid & 7, idx = pid >> 3, q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int mb, nb;
    pk_tile_of(pid, (int)gridDim.x / p.tilesN, p.tilesN, mb, nb);
    const int kt0 = blockIdx.y * p.ktiles_per_split;
    const int nkt = min(p.KB, kt0 + p.ktiles_per_split) - kt0;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const uint32_t lane_off = (uint32_t)lane * 16u;
    // global address (elements) of piece P (0..47) of K step kt.
    //   NT: region P / 16 = A block 0, A block 1, B block; inside a region the packed tile is contiguous (1 KB per piece).
    //   TR: pieces 0..31 = A (feature block P / 4, plane (P % 4) / 2, half P % 2), 32..47 = B likewise.
    auto gaddr = [&](int P, int kt) -> const pk_t* {
        if (TR) {
            const int k = kt0 + kt;
            const bool isb = P >= 32;
            const int q = isb ? P - 32 : P;
            const pk_t* base = isb ? p.B + (size_t)nb * 4 * PK_TILE : p.A + (size_t)mb * 8 * PK_TILE;
            const int fbk = isb ? p.fbB : p.fbA;
            return base + ((size_t)(k >> 2) * fbk + (q >> 2)) * PK_TILE + ((q & 3) >> 1) * PK_IMG + (size_t)(k & 3) * 1024 + (q & 1) * 512;
        }
        const int region = P >> 4, off = P & 15;
        const pk_t* base = region == 2 ? p.B + ((size_t)nb * p.KB + kt0 + kt) * PK_TILE
                                       : p.A + (((size_t)mb * 2 + region) * p.KB + kt0 + kt) * PK_TILE;
        return base + off * 512;
    };
#define WD_DMA(j, kt, stage)                                                                                           \
    do {                                                                                                               \
        const int P_ = wave * 12 + (j);                                                                                \
        const pk_t* g_ = gaddr(P_, kt);                                                                                \
        const uint32_t l_ = (uint32_t)(((stage) * WD_STAGE + P_ * 512) * 2);                                           \
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(g_), "s"(l_) : "memory"); \
    } while (0)
#define WD_DMA_TILE(kt, stage)                                                                                         \
    do {                                                                                                               \
        WD_DMA(0, kt, stage); WD_DMA(1, kt, stage); WD_DMA(2, kt, stage); WD_DMA(3, kt, stage);                        \
        WD_DMA(4, kt, stage); WD_DMA(5, kt, stage); WD_DMA(6, kt, stage); WD_DMA(7, kt, stage);                        \
        WD_DMA(8, kt, stage); WD_DMA(9, kt, stage); WD_DMA(10, kt, stage); WD_DMA(11, kt, stage);                      \
    } while (0)
    const int trq = (lane >> 2) & 3, trchunk = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), trsub = (lane & 1) * 4;
    auto rd = [&](const pk_t* img, int row, int ks) -> pkx8 {
        return *reinterpret_cast<const pkx8*>(img + (row * 4 + ((2 * h + ks) ^ ((row >> 2) & 3))) * 8);
    };
    auto rd_tr = [&](const pk_t* img, int ks) -> pkx8 {
        typedef short short4v __attribute__((ext_vector_type(4)));
        typedef short short8v __attribute__((ext_vector_type(8)));
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const int t0 = 16 * ks + 8 * h + trq, t1 = t0 + 4;
        const pk_t* a0 = img + t0 * 32 + ((trchunk ^ ((t0 >> 2) & 3)) * 8) + trsub;
        const pk_t* a1 = img + t1 * 32 + ((trchunk ^ ((t1 >> 2) & 3)) * 8) + trsub;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a1));
        short8v f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return __builtin_bit_cast(pkx8, f);
    };
    pkx8 f0a[2][4], f0b[2][2], f1a[2][4], f1b[2][2];
    // fragment e (0..11) in order of first use by the rounds (lh x4, hl x4, hh x4): B.h[0,1], A.l[0..3], B.l[0,1], A.h[0..3]
    auto frag_one = [&](int e, const pk_t* s, int ks, pkx8 (&fa)[2][4], pkx8 (&fb)[2][2]) {
        const bool isb = e < 2 || (e >= 6 && e < 8);
        const int pl = isb ? (e < 2 ? 0 : 1) : (e < 6 ? 1 : 0);
        const int i = isb ? (e & 1) : (e < 6 ? e - 2 : e - 8);
        if (TR) {
            if (!isb) fa[pl][i] = rd_tr(s + ((4 * wm + i) * 2 + pl) * 1024, ks);
            else fb[pl][i] = rd_tr(s + 2 * PK_TILE + ((2 * wn + i) * 2 + pl) * 1024, ks);
        } else {
            if (!isb) fa[pl][i] = rd(s + (wm * 2 + pl) * PK_IMG, i * 32 + l31, ks);
            else fb[pl][i] = rd(s + 2 * PK_TILE + pl * PK_IMG, wn * 64 + i * 32 + l31, ks);
        }
    };
    WD_DMA_TILE(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    WD_DMA_TILE(min(1, nkt - 1), 1);
    __builtin_amdgcn_sched_barrier(0);
    WD_DMA_TILE(min(2, nkt - 1), 2);
    __builtin_amdgcn_s_waitcnt(0x4F78);              // vmcnt(24): tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int e = 0; e < 12; ++e) frag_one(e, smem, 0, f0a, f0b);

    auto step = [&](int it, auto tag) {
        // SMODE 1: tiles t+1..t+3 exist; 4: t+1, t+2 exist (nothing more to request); 2: only t+1; 3: last tile
        constexpr int SMODE = decltype(tag)::smode;
        constexpr int CUR = decltype(tag)::cur;              // stage of tile t (0..2)
        constexpr int NXT = (CUR + 1) % 3;
        constexpr bool HAS1 = SMODE != 3, HAS3 = SMODE == 1;
        const pk_t* s_cur = smem + CUR * WD_STAGE;
        const pk_t* s_nxt = smem + NXT * WD_STAGE;
#define WD_MMA(FA, FB, q, i)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa(q)][i], FB[pb(q)][j], acc[i][j], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0)
#define WD_R1(r)                                                                                                \
    frag_one(r, s_cur, 1, f1a, f1b);                                                                             \
    WD_MMA(f0a, f0b, 2 - (r) / 4, (r) & 3)
#define WD_R2(r)                                                                                                \
    if (HAS3) { WD_DMA(r, it + 3, CUR); }                                                                        \
    if (HAS1) { frag_one(r, s_nxt, 0, f0a, f0b); }                                                               \
    WD_MMA(f1a, f1b, 2 - (r) / 4, (r) & 3)
        WD_R1(0); WD_R1(1); WD_R1(2); WD_R1(3); WD_R1(4); WD_R1(5); WD_R1(6); WD_R1(7); WD_R1(8); WD_R1(9); WD_R1(10); WD_R1(11);
        if (SMODE == 1 || SMODE == 4) __builtin_amdgcn_s_waitcnt(0x007C);     // vmcnt(12) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        WD_R2(0); WD_R2(1); WD_R2(2); WD_R2(3); WD_R2(4); WD_R2(5); WD_R2(6); WD_R2(7); WD_R2(8); WD_R2(9); WD_R2(10); WD_R2(11);
#undef WD_MMA
#undef WD_R1
#undef WD_R2
    };
    int it = 0;
    for (; it + 5 < nkt; it += 3) {
        step(it, StepTag<1, 0>{});
        step(it + 1, StepTag<1, 1>{});
        step(it + 2, StepTag<1, 2>{});
    }
    for (; it < nkt; it += 3) {      // tail (it % 3 == 0): 1..5 tiles left
        const int rem = nkt - it;
        if (rem >= 4) step(it, StepTag<1, 0>{}); else if (rem == 3) step(it, StepTag<4, 0>{}); else if (rem == 2) step(it, StepTag<2, 0>{}); else step(it, StepTag<3, 0>{});
        if (rem >= 5) step(it + 1, StepTag<1, 1>{}); else if (rem == 4) step(it + 1, StepTag<4, 1>{}); else if (rem == 3) step(it + 1, StepTag<2, 1>{}); else if (rem == 2) step(it + 1, StepTag<3, 1>{});
        if (rem >= 6) step(it + 2, StepTag<1, 2>{}); else if (rem == 5) step(it + 2, StepTag<4, 2>{}); else if (rem == 4) step(it + 2, StepTag<2, 2>{}); else if (rem == 3) step(it + 2, StepTag<3, 2>{});
    }
#undef WD_DMA
#undef WD_DMA_TILE

    const int flags = p.flags;
    const bool atomic = gridDim.y > 1 && p.split_stride == 0;      // split_stride != 0: split z owns C + z * split_stride
    float* const Cz = p.C + (size_t)blockIdx.y * p.split_stride;
    const float alpha = p.alpha * p.inv_a[0] * p.inv_b[0];
    if (p.epi_f4 && !atomic) {                 // float4 form (host checked alignment and the operand count)
        EpiArgs ea = make_epi_args(Cz, p.bias, p.res, p.relu_src, p.M, p.N, p.ldc, p.ldr, p.ld_relu, flags, 0, alpha, dkn);
        if (p.epi_f4 == 2) epilogue_f4<4, 2, true>(ea, acc, mb * 256 + wm * 128, nb * 128 + wn * 64, lane);
        else epilogue_f4<4, 2, false>(ea, acc, mb * 256 + wm * 128, nb * 128 + wn * 64, lane);
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = nb * 128 + wn * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = (flags & LSTC_EPI_BIAS) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rbase = mb * 256 + wm * 128 + i * 32 + 4 * h;
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
                    v = drop_keep(idx, dkn) ? v * dkn.scale : 0.f;
                }
                if (flags & LSTC_EPI_RESIDUAL) v += p.res[(size_t)row * p.ldr + col];
                if (flags & LSTC_EPI_RELU_MASK) v = p.relu_src[(size_t)row * p.ld_relu + col] > 0.f ? v : 0.f;
                if (flags & LSTC_EPI_ACCUM) v += *cp;
                *cp = v;
            }
        }
    }
}

}  // namespace

// Packed-operand GEMM behind lstc_gemm (dtype LSTC_F32X3): d->A / d->B point to lstc_pack3 outputs for [M, K] / [N, K].
__attribute__((visibility("hidden"))) int lstc_gemm_f32x3_impl(const LstcGemmDesc* d, hipStream_t st) {
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
    PkParams p;
    p.split_stride = splits > 1 ? d->batch_stride_c : 0;      // K splits into separate partial outputs (no atomics)
    p.A = (const pk_t*)d->A; p.B = (const pk_t*)d->B; p.C = (float*)d->C;
    p.bias = d->bias; p.res = (const float*)d->residual; p.relu_src = (const float*)d->relu_src;
    p.M = d->M; p.N = d->N; p.ldc = d->ldc; p.ldr = d->ldr; p.ld_relu = d->ld_relu; p.flags = d->flags; p.alpha = d->alpha;
    p.dk = make_drop_key(d->dropout_p, d->dropout_seed);
    {
        const int naux = ((d->flags & LSTC_EPI_RESIDUAL) ? 1 : 0) + ((d->flags & LSTC_EPI_RELU_MASK) ? 1 : 0) + ((d->flags & LSTC_EPI_ACCUM) ? 1 : 0);
        const bool al = d->N % 4 == 0 && d->N >= 4 && d->ldc % 4 == 0 && aligned16(d->C) && (!(d->flags & LSTC_EPI_BIAS) || aligned16(d->bias)) &&
                        (!(d->flags & LSTC_EPI_RESIDUAL) || (d->ldr % 4 == 0 && aligned16(d->residual))) &&
                        (!(d->flags & LSTC_EPI_RELU_MASK) || (d->ld_relu % 4 == 0 && aligned16(d->relu_src))) && (p.split_stride % 4 == 0);
        p.epi_f4 = (al && naux <= 1) ? (naux ? 2 : 1) : 0;
    }
    // (transA, transB) = (0, 1): A, B are packs of [M, K], [N, K];  (1, 0): packs of the k-major sources [K, M], [K, N]
    const bool tr = d->transA != 0 && d->transB == 0;
    if (!tr && !(d->transA == 0 && d->transB != 0)) return LSTC_E_UNSUPPORTED;
    p.KB = (d->K + 31) / 32;
    p.fbA = (d->M + 31) / 32; p.fbB = (d->N + 31) / 32;
    {   // trailers of the packs: after rows/128 x K/32 tiles of the packed [rows, K] matrix
        const int64_t ra = tr ? d->K : d->M, ka = tr ? d->M : d->K, rb_ = tr ? d->K : d->N, kb_ = tr ? d->N : d->K;
        const int64_t ta = ((ra + 127) / 128) * ((ka + 31) / 32), tb = ((rb_ + 127) / 128) * ((kb_ + 31) / 32);
        p.inv_a = reinterpret_cast<const float*>(p.A + ta * PK_TILE) + 1;
        p.inv_b = reinterpret_cast<const float*>(p.B + tb * PK_TILE) + 1;
    }
    // TR mode streams whole 128-token row blocks of the source packs and 4 feature blocks per 128 outputs
    if (tr && ((d->K % 128) != 0 || (d->M % 128) != 0 || (d->N % 128) != 0)) return LSTC_E_SHAPE;
    p.ktiles_per_split = (p.KB + splits - 1) / splits;
    const int eff_splits = (p.KB + p.ktiles_per_split - 1) / p.ktiles_per_split;
    // (A 256x256-tile form - 16 accumulators per wave, half the L2 traffic - was built and measured: correct, but 3.33 /
    // 5.22 ms against 2.57 / 4.48 ms on the K = 2048 / 4096 forward shapes, its two 64-KB stages leaving 1.5 phases between
    // a request and its use, and its 256 accumulators spilling around the K loop; removed.)
    // 256x128-tile kernel: variant 2, and by default for the weight-gradient form (K = tokens, long loops): 2.01 / 3.68 ms
    // against 2.19 / 4.17 ms of the two-stage 128x128 kernel on 2048x2048 / 4096x2048 outputs over 100352 tokens; on the
    // K = 2048 forward shapes it loses (2.61 vs 2.42 ms: one workgroup per CU, nothing overlaps prologue and epilogue)
    if ((d->variant == 2 && (!tr || (d->M % 256) == 0)) || (d->variant == 0 && tr && (d->M % 256) == 0 && d->K >= 8192)) {
        const int tM = (d->M + 255) / 256;
        p.tilesN = (d->N + 127) / 128;
        constexpr size_t ldsw = (size_t)3 * WD_STAGE * sizeof(pk_t);
        static LstcDevOnce attrw;
        const int devw = attrw.begin();
        if (devw >= 0) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pkw_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pkw_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
            attrw.end(devw);
        }
        if (tr) hipLaunchKernelGGL(gemm_pkw_kernel<true>, dim3(tM * p.tilesN, eff_splits), dim3(NT), ldsw, st, p);
        else hipLaunchKernelGGL(gemm_pkw_kernel<false>, dim3(tM * p.tilesN, eff_splits), dim3(NT), ldsw, st, p);
        return lstc_launch_status();
    }
    const int tilesM = (d->M + 127) / 128;
    p.tilesN = (d->N + 127) / 128;
    // default (variant 0 / 3): the two-stage kernel, two workgroups per CU - 2.44 / 4.30 / 4.93 ms against 2.63 / 4.56 /
    // 5.39 ms of the three-stage one (variant 1) on the 100352 x {2048 x 2048, 2048 x 4096, 4096 x 2048} forward shapes
    if (d->variant != 1) {
        constexpr size_t lds3 = (size_t)2 * PK_STAGE * sizeof(pk_t);
        static LstcDevOnce attr3;
        const int dev3 = attr3.begin();
        if (dev3 >= 0) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pk2s_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pk2s_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            attr3.end(dev3);
        }
        if (tr) hipLaunchKernelGGL(gemm_pk2s_kernel<true>, dim3(tilesM * p.tilesN, eff_splits), dim3(NT), lds3, st, p);
        else hipLaunchKernelGGL(gemm_pk2s_kernel<false>, dim3(tilesM * p.tilesN, eff_splits), dim3(NT), lds3, st, p);
        return lstc_launch_status();
    }
    constexpr size_t lds = (size_t)PK_NSTAGE * PK_STAGE * sizeof(pk_t);
    static LstcDevOnce attr_done;
    const int dev_ = attr_done.begin();
    if (dev_ >= 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pk_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pk_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done.end(dev_);
    }
    if (tr) hipLaunchKernelGGL(gemm_pk_kernel<true>, dim3(tilesM * p.tilesN, eff_splits), dim3(NT), lds, st, p);
    else hipLaunchKernelGGL(gemm_pk_kernel<false>, dim3(tilesM * p.tilesN, eff_splits), dim3(NT), lds, st, p);
    return lstc_launch_status();
}

extern "C" {

int64_t lstc_pack3_bytes(int64_t rows, int64_t K) {
    if (rows <= 0 || K <= 0) return 0;
    // one spare row block when the count is odd: the 256-row kernel may stream it (its products are never stored)
    const int64_t rb = (rows + 127) / 128;
    return (rb + (rb & 1)) * ((K + 31) / 32) * (int64_t)PK_TILE * (int64_t)sizeof(pk_t) + PK_TRAILER;
}

int lstc_pack3(const float* src, int64_t rows, int64_t K, int64_t ld, int32_t k_major, void* dst, void* stream) {
    if (!src || !dst) return LSTC_E_NULL;
    if (rows <= 0 || K <= 0 || ld < (k_major ? rows : K)) return LSTC_E_SHAPE;
    if (!aligned16(dst)) return LSTC_E_ALIGN;
    const int64_t RB = (rows + 127) / 128, KB = (K + 31) / 32;
    if (RB * KB > 0x7fffffffLL || rows > 0x7fffffffLL || K > 0x7fffffffLL) return LSTC_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    uint32_t* trailer = reinterpret_cast<uint32_t*>(reinterpret_cast<pk_t*>(dst) + RB * KB * PK_TILE);
    hipError_t e = hipMemsetAsync(trailer, 0, 16, st);
    if (e != hipSuccess) return (int)e;
    const int64_t srows = k_major ? K : rows, scols = k_major ? rows : K;          // the source as stored
    const int64_t nel = rows * K;
    const unsigned ablocks = (unsigned)(nel >= (1 << 24) ? 1024 : (nel + 16383) / 16384 > 0 ? (nel + 16383) / 16384 : 1);
    hipLaunchKernelGGL(absmax_kernel, dim3(ablocks), dim3(NT), 0, st, src, (long long)srows, (long long)scols, (long long)ld, trailer);
    if (k_major)
        hipLaunchKernelGGL(pack3_km_kernel, dim3((unsigned)(RB * KB)), dim3(NT), 0, st, src, (int)rows, (int)K,
                           (long long)ld, (pk_t*)dst, (int)KB, trailer);
    else
        hipLaunchKernelGGL(pack3_kc_kernel, dim3((unsigned)(RB * ((KB + PK_KPB - 1) / PK_KPB))), dim3(NT), 0, st, src, (int)rows, (int)K,
                           (long long)ld, (pk_t*)dst, (int)KB, trailer);
    return lstc_launch_status();
}

}  // extern "C"
