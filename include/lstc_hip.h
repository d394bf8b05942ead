/*
 * lstc_hip.h — C ABI of liblstc_hip.so, the MI355X (gfx950) native kernels behind the
 * LSTC_VAD training hot path.
 *
 * The reference (shengyangsun/LSTC_VAD) has no FFI / plugin layer: its hot path is the
 * Python class surface models.Encoder / MultiHeadAttention / FFN / Regressor / Classifier
 * plus the loss functions and torch.optim.Adagrad inside the Train/ scripts, all of it implicit
 * ATen kernels.  Each entry point below replaces the ATen work of the reference lines it
 * cites (paths relative to the reference root).  lstc_vad_amd/_lib.py binds them with
 * ctypes; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.
 *  - Every buffer is owned by the caller (PyTorch allocates); the library never allocates
 *    or frees device memory and keeps no pointer after return.
 *  - Asynchronous on the given HIP stream (hipStream_t passed as void*); no internal sync.
 *  - Return 0 on success; <0 = argument/shape error detected on the host before launch
 *    (LSTC_E_*); >0 = hipError_t from the launch.  lstc_strerror() names either.
 *  - All matrices are row-major; "ld" = leading dimension in elements.
 */
#ifndef LSTC_HIP_H
#define LSTC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSTC_VERSION 112            /* 0.1.1 patch 2: see INTEGRATION.md, "ABI history" */

enum {
    LSTC_OK = 0,
    LSTC_E_NULL = -1,               /* required pointer is NULL */
    LSTC_E_SHAPE = -2,              /* non-positive / inconsistent dimension */
    LSTC_E_ALIGN = -3,              /* pointer / leading dimension breaks a documented alignment rule */
    LSTC_E_UNSUPPORTED = -4,        /* combination not implemented */
    LSTC_E_RANGE = -5               /* size exceeds a documented limit (e.g. sequence length, 32-bit indexing) */
};

enum { LSTC_F32 = 0, LSTC_BF16 = 1, LSTC_F32X3 = 2, LSTC_BF16P = 3 };

/* ----------------------------------------------------------------------------- GEMM
 * C[M,N] = epilogue( alpha * op(A)[M,K] * op(B)[K,N] )
 *   transA = 0: A stored [M,K] (K contiguous)      transA = 1: A stored [K,M] (M contiguous)
 *   transB = 0: B stored [K,N] (N contiguous)      transB = 1: B stored [N,K] (K contiguous)
 * Replaces every nn.Linear / torch.matmul-with-weights on the path:
 *   forward  X*W^T  (transA=0, transB=1): models/MultiHeadAttention.py:97-99,123; models/FFN.py:17;
 *                                         models/Regressor.py:19-20; models/Classifier.py:21-22
 *   backward dX = dY*W (0,0) and dW = dY^T*X (1,0): autograd of the above,
 *                                         Train/temporal_transformer_shanghaitech.py:138
 * Epilogue order (each stage optional, selected by `flags`):
 *   v = alpha*acc; v += bias[n]; v = relu(v); v = dropout(v); v += residual[m,n];
 *   v *= (relu_src[m,n] > 0); C = v   (or C += v with LSTC_EPI_ACCUM)
 * fusing  relu(W1 x + b1)            models/FFN.py:17
 *         dropout(W2 h + b2) + x     models/FFN.py:17-19
 *         dropout(fc(o)) + residual  models/MultiHeadAttention.py:123-124
 * Dropout keeps element i = m*N+n iff hash(seed, i) >= p*2^32 and scales kept values by
 * 1/(1-p); lstc_dropout_mask() / lstc_dropout_apply() regenerate the same mask.
 * dtype LSTC_F32: exact f32 MFMA (v_mfma_f32_32x32x2_f32) — bitwise a k-ordered fmaf chain.
 * dtype LSTC_BF16: A/B bf16, f32 accumulate; C bf16 unless LSTC_EPI_OUT_F32.
 * dtype LSTC_F32X3: f32-accurate product on the 16-bit matrix cores (3 f16 plane products per f32 product); A and B are PACKED operands produced by lstc_pack3
 *   (lda/ldb ignored).  (transA, transB) = (0, 1): packs of the [M,K] and [N,K] matrices (lstc_pack3 transposes k-major
 *   sources on the way).  (1, 0): packs of the k-major SOURCES [K,M] and [K,N] themselves - the weight-gradient product
 *   dY^T X reuses the packs the forward / input-gradient products made of X and dY; needs M, N, K multiples of 128.
 *   C, bias, residual, relu_src and the epilogue are f32 exactly as for LSTC_F32; no batch.
 * dtype LSTC_BF16P: the LSTC_BF16 arithmetic (operands rounded to bf16 RNE, f32 accumulate, f32 C / epilogue) on PACKED bf16
 *   operands produced by lstc_pack1 (lda/ldb ignored): 2 B per element through L2/LDS instead of 4, streamed by LDS-DMA into
 *   a 256x256x64-tile kernel (csrc/gemm_bf16p.hip).  (0, 1): packs of [M,K] and [N,K];  (1, 0): packs of the k-major SOURCES
 *   [K,M] and [K,N] (weight gradient dY^T X reusing the packs of dY and X; K must be a multiple of 128).  No batch.
 */
enum {
    LSTC_EPI_BIAS = 1, LSTC_EPI_RELU = 2, LSTC_EPI_DROPOUT = 4, LSTC_EPI_RESIDUAL = 8,
    LSTC_EPI_RELU_MASK = 16, LSTC_EPI_ACCUM = 32, LSTC_EPI_OUT_F32 = 64,
    /* LSTC_BF16P only, (0, 1) form, M and N multiples of 256, no K split (else LSTC_E_UNSUPPORTED):
     *   OUT_PACK:       C is an lstc_pack1 buffer (lstc_pack1_bytes(M, N)) that receives the result rounded to bf16 in the layout
     *                   of a packed [M, N] operand - the output of W1 (models/FFN.py:17) IS the A operand of W2 and of dW2, and
     *                   the gradient of that hidden IS the operand of dW1 / dX: no f32 copy, no lstc_pack1 pass (ldc ignored);
     *   RELU_MASK_PACK: relu_src is such a pack (the hidden's sign is read from its bf16 form; ld_relu ignored);
     *   RESIDUAL_PACK:  residual is such a pack of [M, N] (ldr ignored; with RESIDUAL and OUT_PACK, never with RELU_MASK): the
     *                   bf16 activation stream - dropout(fc(o)) + x (models/MultiHeadAttention.py:123-124), dropout(W2 h + b2) + x
     *                   (models/FFN.py:17-19) and the input gradients dX + dy read their residual as 2 bytes per element from
     *                   the pack the previous block's LayerNorm wrote and leave 2 bytes per element for the next one. */
    LSTC_EPI_OUT_PACK = 128, LSTC_EPI_RELU_MASK_PACK = 256, LSTC_EPI_RESIDUAL_PACK = 512
};

#define LSTC_VARIANT_NO_QTAIL (1 << 30)

typedef struct LstcGemmDesc {
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int32_t transA, transB;
    int32_t dtype;                  /* LSTC_F32 | LSTC_BF16 */
    int32_t flags;                  /* LSTC_EPI_* */
    float   alpha;
    float   dropout_p;
    uint64_t dropout_seed;
    int32_t ldr;                    /* residual leading dimension */
    int32_t ld_relu;                /* relu_src leading dimension */
    int32_t split_k;                /* 0/1 = none; >1: K split over workgroups, partial sums added with f32 atomics
                                       into C, which the caller must have zeroed (only alpha epilogue allowed).
                                       LSTC_F32X3 with batch_stride_c != 0: split z writes its partial product to
                                       C + z*batch_stride_c instead (no atomics; the caller sums the partials) */
    int32_t variant;                /* 0 = library default tile (LSTC_F32: 128x128 tiles, one tile per workgroup, with the rows of a
                                       mostly empty last tile round on the 64x64 variant - bit-identical results; the default NEVER
                                       selects the persistent kernel); LSTC_F32 in the production library: 4 = the default loop
                                       without the row split, 8 = its fallback for unaligned operands (scalar epilogue: the
                                       reference of the bitwise tests), 11 = the 64x64 tail tile, 12 = the persistent walk of the
                                       default loop (bit-identical, measured slower); anything else -> LSTC_E_UNSUPPORTED.  The
                                       other tile variants (1-3, 5-7, 9, 10) and the timing-only ablations (13-15) exist only in
                                       -DLSTC_TUNING builds (`make tuning`, tools/tuning/gemm_check), never in the production library.
                                       LSTC_BF16P: 0, or LSTC_VARIANT_NO_QTAIL = the product as ONE persistent launch of 256 x 256 tiles,
                                       without the quarter-tile kernel that otherwise takes the tiles behind the last whole round of
                                       workgroups (bit-identical results: the reference of the bitwise tests) */
    int32_t batch;                  /* 0/1 = single problem; >1: `batch` independent problems of identical shape, problem z
                                       uses A + z*batch_stride_a etc. (elements).  Per-head products of the last layer's
                                       CLS attention (q_h W_k,h etc.).  Only alpha / ACCUM epilogues. */
    int64_t batch_stride_a, batch_stride_b, batch_stride_c;
    const void* A;
    const void* B;
    void*       C;
    const float* bias;              /* [N] */
    const void*  residual;          /* [M,ldr], same dtype as C */
    const void*  relu_src;          /* [M,ld_relu], same dtype as C */
} LstcGemmDesc;

int lstc_gemm(const LstcGemmDesc* d, void* stream);

/* Number of K slices lstc_gemm actually launches for (dtype, K, split_k): ceil(kt / ceil(kt / split_k)) with kt = K tiles
 * of that dtype's kernel.  It can be SMALLER than split_k (e.g. 132 K tiles, split_k 16 -> 15 slices of 9 tiles), so a
 * caller that gives each slice its own partial output (batch_stride_c) must size and sum exactly this many. */
int32_t lstc_gemm_splits(int32_t dtype, int32_t K, int32_t split_k);

/* Operand packing for LSTC_F32X3.  The tensor is scaled by the power of two s that puts its largest magnitude in
 * [2^14, 2^15) and written as 128-row x 32-k tiles (row = an output row of A's side or an output column of B's side, K = the
 * contraction); each tile = two 8-KB f16 planes h = f16(x s), l = f16(x s - h), stored in the LDS image layout the GEMM
 * streams with global_load_lds (csrc/gemm_pk.hip); rows and K are zero-padded to the tile; a 256-B trailer carries the
 * tensor's absmax bits and 1/s (the GEMM epilogue multiplies by 1/(s_a s_b)).  Three launches: memset, absmax, pack.
 *   k_major = 0: src is [rows, K] with K contiguous (ld >= K)  - X of X*W^T, W of X*W^T, dY of dY*W
 *   k_major = 1: src is [K, rows] with rows contiguous (ld >= rows) - W of dY*W (and dY, X of dY^T*X when the
 *                (1, 0) form of lstc_gemm is not applicable)
 * dst needs lstc_pack3_bytes(rows, K) bytes, 16-B aligned.  The same nn.Linear products as lstc_gemm (see above). */
int64_t lstc_pack3_bytes(int64_t rows, int64_t K);
int lstc_pack3(const float* src, int64_t rows, int64_t K, int64_t ld, int32_t k_major, void* dst, void* stream);

/* Operand packing for LSTC_BF16P: the logical [rows, K] operand rounded to bf16 (RNE) and stored as 128-row x 32-k tiles,
 * each tile the 8-KB LDS image the GEMM reads (64-B rows, 16-B chunk index XOR (row >> 2) & 3); rows and K are zero-padded
 * to an EVEN number of tiles each way.  k_major as for lstc_pack3.  One launch.  dst needs lstc_pack1_bytes(rows, K) bytes,
 * 16-B aligned.  Same nn.Linear products as lstc_gemm (models/MultiHeadAttention.py:97-99,123; models/FFN.py:17). */
int64_t lstc_pack1_bytes(int64_t rows, int64_t K);
int lstc_pack1(const float* src, int64_t rows, int64_t K, int64_t ld, int32_t k_major, void* dst, void* stream);
/* Several operands in ONE launch - the weights of a model after an optimizer step (models/MultiHeadAttention.py:97-99,123 and
 * models/FFN.py:17,19 each need their nn.Linear weight in both layouts: ~35 launches of 10-20 us otherwise).  Each item as
 * lstc_pack1's arguments; results are bit for bit those of `count` lstc_pack1 calls. */
typedef struct LstcPackItem {
    const float* src; int64_t rows, K, ld; int32_t k_major; void* dst;
} LstcPackItem;
int lstc_pack1_multi(const LstcPackItem* items, int32_t count, void* stream);
/* out[k] (+)= sum over rows of the packed [rows, K] operand (bf16 values added in f32; two passes through `partial`
 * [n_partial, K]): the bias gradient db1 = column sums of the hidden's gradient (autograd of models/FFN.py:17) when that
 * gradient exists only as the packed output of a LSTC_EPI_OUT_PACK product. */
int lstc_colsum_pack1(const void* packed, int64_t rows, int32_t K, float* partial, int32_t n_partial, float* out,
                      int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------ attention
 * Fused core of models/MultiHeadAttention.py:103-122 for one layer, all sequences, heads:
 *   A = (Q/sqrt(d_k)) K^T ; A[:, :, 1:, 1:] += table[index[i-1, j-1], h] ; P = softmax(A, -1) ;
 *   Pd = dropout(P) ; O = Pd V, written head-merged ([N, S, H*dv]).
 * Q, K, V are the projection outputs as the GEMM leaves them: [N, S, H*dk] (token-major, heads
 * interleaved) — the reference's transpose(1,2)/contiguous copies (:101,:122) disappear.
 * `probs` [N, H, S, S] receives P (pre-dropout) for the backward pass / return_attn.
 * rel-bias: `table` [n_rows, H] float, `index` int64 [index_ld-wide rows]; entry used for
 * (i, j), 1 <= i,j < S is table[index[(i-1)*index_ld + (j-1)], h] — the top-left (S-1)x(S-1)
 * block, exactly the reference's `relative_position_index[:len_q-1, :len_q-1]` slice (:108).
 */
typedef struct LstcAttnDesc {
    int32_t N, S, H, dk, dv;        /* dk, dv multiples of 32; S <= 128 */
    int32_t ldq, ldk, ldv, ldo;     /* token strides in elements (normally H*dk / H*dv) */
    int32_t dtype;                  /* LSTC_F32: every product on the exact-f32 MFMA.  LSTC_BF16 (bf16 training mode): the operands of
                                       Q K^T, Pd V and of the four backward products are rounded to bf16 (RNE) in registers and
                                       contracted by v_mfma_f32_32x32x16_bf16 with f32 accumulation; Q, K, V, O, probs and the
                                       gradients stay f32 in memory, softmax / bias / dropout stay f32.  Taken by the staged
                                       kernels (S <= 96, dk and dv multiples of 32, 16-B aligned operands, variant 0); every other
                                       case computes the exact-f32 products (the first-generation loops are latency-bound and got
                                       slower with bf16 products).  lstc_attn_cls_* : LSTC_F32 only */
    int32_t index_ld;               /* 0 = no relative bias */
    int32_t table_rows;             /* rows of `table` / `dtable` (backward: size of the per-workgroup LDS accumulator) */
    float   scale;                  /* 1/sqrt(d_k): multiplies Q (reference divides by temperature :49,:103) */
    float   dropout_p;
    uint64_t dropout_seed;
    const void* Q; const void* K; const void* V;
    void*  O;
    float* probs;                   /* [N,H,S,S] f32 */
    const float*   table;           /* [rows, H] or NULL */
    const int64_t* index;           /* or NULL */
    /* backward only */
    const void* dO;                 /* [N,S,H*dv] */
    void* dQ; void* dK; void* dV;   /* [N,S,H*dk|dv], written (not accumulated) */
    float* dtable;                  /* dtable_chunks == 0: [rows,H] accumulated with atomics, caller zeroes; NULL = skip.
                                       dtable_chunks > 0: [chunks][rows][H] partial tables, one per chunk of
                                       ceil(N/chunks) sequences, written (no atomics; the caller sums them - a
                                       fixed-order, bit-reproducible gradient) */
    int32_t dtable_chunks;
    int32_t variant;                /* 0 = default kernels; 1 = first-generation kernels (operands straight from global in
                                       MFMA lane layout) - kept for A/B measurements and as the fallback for shapes the
                                       staged kernels do not take (d_k or d_v not a multiple of 32, unaligned operands, S > 96);
                                       forward only: 2 = the second-generation (LDS-logit) kernel where the default is the
                                       lane-=-query f32 kernel (LSTC_F32, S <= 32 or 64 < S <= 96, d_v a multiple of 64),
                                       3 = that kernel also for 32 < S <= 64 (A/B measurements; same results to rounding).
                                       Packed-input forward only (in_pack_cols > 0): 100 + n = n sequences per workgroup
                                       (1 <= n <= 16; the default picks n from N * H), a measurement hook - results do not
                                       depend on it */
    /* backward, bf16 mode: when all three are non-NULL, dQ / dK / dV are written ONLY as packed bf16 operands [N*S, H*dk|dv]
     * (lstc_pack1 layout, lstc_pack1_bytes(N*S, H*dk|dv) bytes each; dQ / dK / dV may be NULL) - they feed the packed weight- and
     * input-gradient products of the projections (autograd of models/MultiHeadAttention.py:97-99).  Needs the staged kernel
     * (S <= 96, d_k, d_v multiples of 32), N*S a multiple of 256, H*dk and H*dv multiples of 64; else LSTC_E_UNSUPPORTED. */
    void* dQ_pack; void* dK_pack; void* dV_pack;
    /* pack_cols > 0: the three pointers name ONE pack of a [N*S, pack_cols] matrix (the fused Q|K|V projection's gradient) whose
     * columns dQ_col0 / dK_col0 / dV_col0 .. + H*dk|dv receive dQ / dK / dV (multiples of 32, pack_cols a multiple of 64). */
    int32_t pack_cols, dQ_col0, dK_col0, dV_col0;
    /* forward, bf16 mode: when non-NULL, O is written ONLY as a packed bf16 operand [N*S, H*dv] (O may be NULL) - it feeds the
     * packed products of the output projection fc (models/MultiHeadAttention.py:122-123) and of its weight gradient.  N*S a
     * multiple of 256, H*dv a multiple of 64, d_v a multiple of 32; else LSTC_E_UNSUPPORTED. */
    void* O_pack;
    /* bf16 mode, PACKED INPUTS (third-generation kernels): in_pack_cols > 0 says that Q, K, V point at lstc_pack1 buffers of
     * a [N*S, in_pack_cols] matrix - normally all three at ONE pack, the fused Q|K|V projection written with LSTC_EPI_OUT_PACK
     * (models/MultiHeadAttention.py:97-99) - whose columns Q_col0 / K_col0 / V_col0 .. + H*dk|dv hold the projections (multiples
     * of 32; in_pack_cols a multiple of 64); ldq / ldk / ldv are ignored.  Backward: dO_pack_cols > 0 likewise makes dO a pack of
     * [N*S, dO_pack_cols], columns dO_col0 .. + H*dv (the packed input gradient of fc, :123).  Packed inputs come with packed
     * outputs only: the forward needs O_pack, the backward dQ_pack / dK_pack / dV_pack and both packed inputs.  dtype LSTC_BF16,
     * S <= 96, d_k a multiple of 32, d_v of 64, N*S of 256; else LSTC_E_UNSUPPORTED.  No f32 copy of Q, K, V, O, dO, dQ, dK or dV
     * exists in this form: the attention core moves 2 bytes per element each way. */
    int32_t in_pack_cols, Q_col0, K_col0, V_col0;
    int32_t dO_pack_cols, dO_col0;
    /* row pitch of `probs` in floats: 0 or S = dense [N,H,S,S].  The packed-input kernels need a multiple of 4 (>= S; `probs`
     * 16-B aligned): they move the probabilities as 16-B groups, the forward writes zeros into the padding columns and the
     * backward expects them there. */
    int32_t probs_ld;
} LstcAttnDesc;

int lstc_attn_fwd(const LstcAttnDesc* d, void* stream);
/* autograd of the above (dV = Pd^T dO; dPd = dO V^T; dA = P*(dP - rowsum(dP*P)); dQ = dA K*scale;
 * dK = dA^T Q*scale; dtable[index] += dA[1:,1:] summed over sequences). */
int lstc_attn_bwd(const LstcAttnDesc* d, void* stream);

/* CLS-query attention for the LAST encoder layer: only enc_output[:, 0, :] is consumed downstream
 * (Train/temporal_transformer_shanghaitech.py:123; Train/spatio_transformer_shanghaitech.py:97), so its queries are
 * needed for token 0 only while keys/values still span all S tokens (row 0 carries no relative bias:
 * models/MultiHeadAttention.py:111 adds it to attn[:, :, 1:, 1:]).  Same descriptor; Q/dQ/O/dO hold ONE row per
 * sequence ([N, ldq] / [N, ldo]) and probs is [N, H, S]. */
int lstc_attn_cls_fwd(const LstcAttnDesc* d, void* stream);
int lstc_attn_cls_bwd(const LstcAttnDesc* d, void* stream);

/* Re-associated CLS attention of the last layer: with one query per (sequence, head) the key / value projections are
 * never materialised — score[n,h,j] = u[n,h].x[n,j] with u = (q_h*scale) Wk_h, and o[n,h] = Wv_h (sum_j p[n,h,j] x[n,j]) —
 * which removes the layer's two [N*S,d]x[d,H*dk] projection GEMMs and their four backward GEMMs (same reference lines
 * as above: models/MultiHeadAttention.py:97-122 restricted to row 0).  U, Y: [N,H,d]; X, dX: [N,S,d]; W*, out, probs: [N,H,S].
 *   lstc_cls_dot   out[n,h,j] = sum_c U[n,h,c] X[n,j,c];  mode 0 raw | 1 softmax (+dropout), probs saved |
 *                  2 softmax/dropout backward: out = P*(dP - sum dP*P), dP = dot*keep
 *   lstc_cls_wsum  Y[n,h,c]  = sum_j W[n,h,j] X[n,j,c]
 *   lstc_cls_outer dX[n,j,c] = sum_h W1[n,h,j] U1[n,h,c] + W2[n,h,j] U2[n,h,c]
 * Requirements: d % 4 == 0, 16-B aligned pointers, S <= 128, H <= 16. */
int lstc_cls_dot(const float* U, const float* X, float* out, float* probs, int64_t N, int32_t S, int32_t H, int32_t d,
                 int32_t mode, float dropout_p, uint64_t seed, void* stream);
int lstc_cls_wsum(const float* W, const float* X, float* Y, int64_t N, int32_t S, int32_t H, int32_t d, void* stream);
int lstc_cls_outer(const float* W1, const float* U1, const float* W2, const float* U2, float* dX, int64_t N, int32_t S,
                   int32_t H, int32_t d, void* stream);

/* The three passes above over a PACKED X (round 5: the bf16 activation stream hands the last full layer's output to the CLS-only
 * layer as an lstc_pack1 operand of the [N*S, d] matrix and takes the gradient back as one; same reference lines).  Arithmetic is
 * f32 on the widened bf16 values; lstc_cls_outer_pack rounds dX once (RNE) after adding `add0` [N, d] (may be NULL) to row 0 of
 * every sequence - the CLS row's own terms dQ Wq + the residual gradient.  U, Y, W*, out, probs as above.
 * Requirements: (N*S) % 256 == 0, d = 512 / 1024 / 2048, H <= 8 (LSTC_E_UNSUPPORTED otherwise), S <= 128, 16-B aligned pointers. */
int lstc_cls_dot_pack(const float* U, const void* X_pack, float* out, float* probs, int64_t N, int32_t S, int32_t H, int32_t d,
                      int32_t mode, float dropout_p, uint64_t seed, void* stream);
int lstc_cls_wsum_pack(const float* W, const void* X_pack, float* Y, int64_t N, int32_t S, int32_t H, int32_t d, void* stream);
int lstc_cls_outer_pack(const float* W1, const float* U1, const float* W2, const float* U2, const float* add0, void* dX_pack,
                        int64_t N, int32_t S, int32_t H, int32_t d, void* stream);

/* ------------------------------------------------------------------- row-wise kernels */
/* y = LayerNorm(x) * gamma + beta over the last dim (eps inside the sqrt, biased variance) —
 * nn.LayerNorm(d_model, eps=1e-6): models/MultiHeadAttention.py:47,125-126; models/FFN.py:10,20-21;
 * models/Encoder.py:31,48-49.  Saves mean / rstd per row for the backward. */
int lstc_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int64_t rows, int32_t d, float eps, void* stream);
/* dx, and per-workgroup partial dgamma/dbeta in `partial` [2, n_partial, d] (reduce with lstc_colsum). */
int lstc_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx, float* partial, int32_t n_partial, int64_t rows, int32_t d, void* stream);
/* bf16 mode: the same two kernels also emit the packed bf16 operand (lstc_pack1 layout, [rows, d]) of the GEMMs that read
 * their result, from the registers that hold the row - the separate lstc_pack1 pass (4 B read + 2 B written per element)
 * and, in the backward, the lstc_dropout_apply pass disappear.
 *   lstc_layernorm_fwd_pack:      `packed` = pack of y: the A operand of the next block's first product (the Q/K/V
 *                                 projections after models/FFN.py:20-21, W1 after models/MultiHeadAttention.py:125-126).
 *   lstc_layernorm_bwd_drop_pack: `packed` = pack of df = dropout-replay(dx) (keep iff the hash of the flat index row*d+col
 *                                 passes, scaled by 1/(1-p) - exactly lstc_dropout_apply(dx, p, seed)): the operand of the
 *                                 weight gradient and the input gradient of the Linear in front of the dropout
 *                                 (fc: MultiHeadAttention.py:123, w_2: FFN.py:17-18); `partial` is [3, n_partial, d], the
 *                                 third plane = column sums of df (that Linear's bias gradient).  dx stays f32 (residual).
 * Both need rows % 256 == 0, d % 64 == 0, d <= 2048 (the rows fill the pack's even tile grid exactly; LSTC_E_UNSUPPORTED
 * otherwise - the caller then packs with lstc_pack1) and `packed` of lstc_pack1_bytes(rows, d) bytes. */
int lstc_layernorm_fwd_pack(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            int64_t rows, int32_t d, float eps, void* packed, void* stream);
/* The same fusion with an f32 result (fp32 / f32x3 modes): df = lstc_dropout_apply(dx, p, seed) written next to dx, `partial`
 * [3, n_partial, d] with the column sums of df as third plane.  d = 512, 1024 or 2048 (LSTC_E_UNSUPPORTED otherwise). */
int lstc_layernorm_bwd_drop(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                            float* dx, float* df, float* partial, int32_t n_partial, int64_t rows, int32_t d, float dropout_p,
                            uint64_t dropout_seed, void* stream);
int lstc_layernorm_bwd_drop_pack(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 float* dx, float* partial, int32_t n_partial, int64_t rows, int32_t d, float dropout_p,
                                 uint64_t dropout_seed, void* packed, void* stream);

/* bf16 ACTIVATION STREAM (round 5; the bf16 compute mode's default where the shapes allow).  Between the CLS concat and the last
 * full encoder layer no f32 activation exists: the residual sums dropout(f) + x (models/MultiHeadAttention.py:123-124,
 * models/FFN.py:17-19) leave the GEMM epilogues as lstc_pack1 operands (LSTC_EPI_OUT_PACK + LSTC_EPI_RESIDUAL_PACK), these two
 * kernels run nn.LayerNorm (models/MultiHeadAttention.py:125-126, models/FFN.py:20-21) and its backward ON the packs, and the
 * gradient of the residual stream travels as packs too - 2 bytes per element per pass instead of 4 (+2 for the pack).
 * Statistics, normalisation, dropout replay and the partial sums are f32 arithmetic; only what is stored is bf16 (RNE).
 *   lstc_layernorm_fwd_act: exactly one of x (f32 [rows, d]) / x_pack (pack of [rows, d]) is the input; y (f32) and / or y_pack
 *                           receive the result (f32: the stream's exit where the CLS-only layer cannot read a pack).
 *   lstc_layernorm_bwd_act: x_pack = the forward's input pack; the incoming gradient is dy (f32) or dy_pack; dx_pack (may be
 *                           NULL: nobody reads layer 0's) = gradient of the residual sum, df_pack = dropout-replay(dx) exactly
 *                           as lstc_layernorm_bwd_drop_pack, partial [3, n_partial, d] = dgamma, dbeta, column sums of df.
 * rows % 256 == 0 and d = 1024 or 2048 (LSTC_E_UNSUPPORTED otherwise); every pack lstc_pack1_bytes(rows, d) bytes, 16-B aligned. */
int lstc_layernorm_fwd_act(const float* x, const void* x_pack, const float* gamma, const float* beta, float* y, void* y_pack,
                           float* mean, float* rstd, int64_t rows, int32_t d, float eps, void* stream);
int lstc_layernorm_bwd_act(const float* dy, const void* dy_pack, const void* x_pack, const float* gamma, const float* mean,
                           const float* rstd, void* dx_pack, float* partial, int32_t n_partial, int64_t rows, int32_t d,
                           float dropout_p, uint64_t dropout_seed, void* df_pack, void* stream);

/* CLS = mean over tokens (or `cls_token` if not NULL), prepended; optional `pos` [S, d] added to every
 * sequence — models/Encoder.py:51-58.  x [N, S-1, d] -> y [N, S, d].  When `x_hi` is not NULL, sequences
 * [0, n_lo) are read from `x` and [n_lo, N) from `x_hi`: the reference's torch.cat([norm_feats, abnorm_feats])
 * (Train/temporal_transformer_shanghaitech.py:120) is fused into this pass instead of costing its own copy. */
int lstc_cls_concat_fwd(const float* x, const float* x_hi, int64_t n_lo, const float* cls_token, const float* pos,
                        float* y, int64_t N, int32_t S, int32_t d, void* stream);
/* The same with the packed bf16 form of y ([N*S, d], lstc_pack1 layout) written next to it: layer 0's A operand in bf16 mode
 * (N*S a multiple of 256, d a multiple of 64; else LSTC_E_UNSUPPORTED).  y may be NULL (round 5, the bf16 activation stream: the
 * pack is then the only output; d % 4 == 0 and 16-B aligned operands required). */
int lstc_cls_concat_fwd_pack(const float* x, const float* x_hi, int64_t n_lo, const float* cls_token, const float* pos,
                             float* y, int64_t N, int32_t S, int32_t d, void* packed, void* stream);

/* The same pass fed STRAIGHT from an HBM-resident feature bank [bank_clips, P, d] (round 5): token t of sequence n is patch t % P of
 * clip clip_idx[n * (S-1)/P + t / P], i.e. lstc_gather_rows (the batch formation `feat[chosen[...], :]`, utils/load_dataset.py:88 +
 * default collate) + the torch.cat + the CLS concat in ONE pass over the features - the gathered batch [B, T, P, d] is never
 * written.  y (f32 [N, S, d]) and / or packed (lstc_pack1 of [N*S, d]: N*S a multiple of 256, d of 64) receive the result.
 * (S-1) % P == 0, d % 4 == 0, 16-B aligned bases; idx entries must lie in [0, bank_clips). */
int lstc_cls_concat_gather_fwd(const float* bank, int64_t bank_clips, const int64_t* clip_idx, int32_t P, const float* cls_token,
                               const float* pos, float* y, int64_t N, int32_t S, int32_t d, void* packed, void* stream);

/* Gradient w.r.t. the Encoder input, needed only when something upstream is trainable (input_layerNorm,
 * models/Encoder.py:48-49): dx[n,t,:] = dy[n,t+1,:] + (mean_cls ? dy[n,0,:]/(S-1) : 0). */
int lstc_cls_concat_bwd(const float* dy, float* dx, int64_t N, int32_t S, int32_t d, int32_t mean_cls, void* stream);

/* out[c] = sum_r x[r, c] for x [rows, ld] (first `cols` columns); deterministic two-pass reduction.
 * `partial` is caller workspace of n_partial*cols floats.  Bias / LayerNorm / pos-enc gradients. */
int lstc_colsum(const float* x, int64_t rows, int32_t cols, int32_t ld, float* partial, int32_t n_partial,
                float* out, int32_t accumulate, void* stream);
/* `batch` column sums in the same two launches: plane b is x + b * batch_stride ([rows, ld]), its result out[b * cols ...];
 * `partial` holds batch * min(rows, n_partial) * cols floats.  Per plane the arithmetic (row -> partial row -> accumulator, order
 * of every addition) is lstc_colsum's two-pass form, so out[b] is bit-identical to lstc_colsum of plane b.  The LayerNorm
 * backward's three per-workgroup partial planes (dgamma, dbeta, dbias: autograd of nn.LayerNorm / nn.Linear.bias, models/FFN.py:
 * 19-21, models/MultiHeadAttention.py:123-126) are reduced by ONE call instead of three. */
int lstc_colsum_batched(const float* x, int32_t batch, int64_t rows, int32_t cols, int32_t ld, int64_t batch_stride,
                        float* partial, int32_t n_partial, float* out, void* stream);

/* y[i] = x[i] * keep(i)/(1-p) with the GEMM epilogue's mask (dropout backward / standalone dropout). */
int lstc_dropout_apply(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);
/* The same on an lstc_pack1 operand [rows, d] (rows % 256 == 0, d % 64 == 0): y_pack = dropout-replay(x_pack) with the mask of
 * the flat index row * d + col, bf16 in, bf16 out (one more RNE rounding).  The bf16 activation stream's backward of a block
 * WITHOUT LayerNorm (models/MultiHeadAttention.py:123-124 with layerNorm = False: the STN configs): the incoming gradient pack
 * is the gradient of dropout(f) + x, its dropped form the operand of fc's weight and input gradients. */
int lstc_dropout_apply_pack(const void* x_pack, void* y_pack, int64_t rows, int32_t d, float p, uint64_t seed, void* stream);
/* Second half of a K-split small product (round 5): C = epi(parts[0] + ... + parts[splits - 1]), parts[i] = the [M, N] partial result
 * of K chunk i at parts + i * part_stride (row-major, ld = N), added in chunk order; `flags` and the operands as LstcGemmDesc's
 * epilogue (bias, ReLU, dropout of the flat index row * N + col, residual, ReLU mask, accumulate - in that order; no pack flags).
 * The CLS-only last layer and the heads run [sequences, d] x [d, d] products whose few output tiles leave most of the chip idle
 * while one workgroup walks the whole K range (models/MultiHeadAttention.py:97-126 on row 0, models/FFN.py:17-19,
 * models/Classifier.py:8-14 for a rank's 256 sequences): the host side launches their K chunks as ONE batched lstc_gemm into
 * `parts` and finishes here.  `groups` > 1: that many independent [M, N] results in one call (the per-head products q_h W_k,h /
 * xbar_h W_v,h^T of the re-associated CLS attention) - group g's partials at parts + g * group_stride_parts, its result at
 * C + g * group_stride_c; only the plain sum (optionally LSTC_EPI_ACCUM).  groups = 1: the strides are ignored.
 * N, part_stride, ldc (ldr, ld_relu, group strides) multiples of 4; 16-B aligned pointers; groups * M * N < 2^32. */
int lstc_splitk_finish(const float* parts, int32_t splits, int64_t part_stride, int64_t M, int64_t N, const float* bias,
                       const float* residual, int64_t ldr, const float* relu_src, int64_t ld_relu, float* C, int64_t ldc, int32_t flags,
                       float dropout_p, uint64_t dropout_seed, int32_t groups, int64_t group_stride_parts, int64_t group_stride_c,
                       void* stream);
/* out[i, 0:K] = the bf16 values of row row0 + i * row_step of an lstc_pack1 operand [rows, K], widened to f32 (i < n; K % 8 == 0,
 * ldo % 4 == 0).  The CLS-only last layer reads its query rows (token 0 of every sequence: row0 = 0, row_step = S) out of the
 * activation stream's pack (models/MultiHeadAttention.py:97 restricted to row 0); tests read whole packs back with it. */
int lstc_unpack1_rows(const void* x_pack, int64_t rows, int32_t K, int64_t row0, int64_t row_step, int64_t n, float* out, int64_t ldo,
                      void* stream);
/* mask[i] = keep(i) ? 1 : 0 — exported so tests can replay a HIP dropout run through the oracle. */
int lstc_dropout_mask(uint8_t* mask, int64_t n, float p, uint64_t seed, void* stream);
/* Dropout seeds for a captured step (hipGraph).  Every entry that draws a dropout mask takes its 64-bit seed BY VALUE, so a
 * captured launch would replay one mask for ever.  While `dev_word` is non-NULL, every such launch of this process (from any host
 * thread - the autograd backward runs on its own) carries the pointer and uses  seed + *dev_word,  read ON THE DEVICE when the
 * kernel runs: the caller bumps the word between replays (by the number of seeds a step draws) and the replayed step sees the
 * masks the eager step with those seeds would have seen, bit for bit.  NULL (the default) restores plain by-value seeds.
 * Process-wide launch context, meant to bracket a stream capture; the word must outlive every graph that captured it.
 * Replaces nothing upstream: the reference draws its masks from torch's global generator (nn.Dropout, models/FFN.py:12). */
int lstc_dropout_seed_device(const uint64_t* dev_word);

/* Last head layer fused with its activation: out = sigmoid(x W^T + b) (c=1, models/Regressor.py:9) or
 * softmax(x W^T + b) (c=2, models/Classifier.py:10).  x [rows, 32]. */
int lstc_head_out_fwd(const float* x, const float* W, const float* b, float* out, int64_t rows, int32_t c,
                      void* stream);
/* dx [rows,32], dW [c,32], db [c] from d(out); dW/db are WRITTEN (round 4: no caller-side fill) by one workgroup that sums in a fixed order. */
int lstc_head_out_bwd(const float* x, const float* W, const float* out, const float* dout,
                      float* dx, float* dW, float* db, int64_t rows, int32_t c, void* stream);

/* ----------------------------------------------------------------------------- loss
 * MIL ranking loss + sparsity (+ CE on softmax outputs | + weighted BCE) and its gradient w.r.t. the
 * head output, one launch:
 *   mode 0 (STN):  Train/spatio_transformer_shanghaitech.py:21-32
 *   mode 1 (LTN):  Train/temporal_transformer_shanghaitech.py:21-36,125-134  (out [2bs*pn, 2]; score = out[:,1])
 *   mode 2 (STN + BCE): Train/spatio_transformer_MIL_CE.py:23-26,32-44,176-181
 * bag[v] = max_p mean_l score[v,p,l]; err = sum_ij relu(1 - abn_j + nor_i)/bs^2;
 * l1 = mean(score[l1_skip:]) (flat slice quirk: l1_skip = bs*pn*L for STN, bs for LTN / co-teach).
 * Soft targets are built in-kernel (Train/temporal_transformer_shanghaitech.py:103-112): normal parts
 * [1,0]; abnormal parts t1 = mean over label_len pseudo labels, t0 = 1-t1.
 * Data-parallel: the hinge couples all bs_global x bs_global pairs, so ranks exchange bag maxima
 * (2*bs_global floats, one sum-all-reduce) between phase 0 and phase 1; every other term is a local sum
 * divided by a global count.  scalars = rank-local contributions (their sum over ranks is the reference
 * value; on one GPU they are the reference values).
 */
typedef struct LstcLossDesc {
    int32_t mode;                   /* 0 STN, 1 LTN (MIL + CE), 2 STN co-teaching (MIL + BCE) */
    int32_t bs_global, bs_local, rank_off;  /* pairs per global batch / on this rank / first pair index of this rank */
    int32_t part_num;
    int32_t score_len;              /* scores per part in `out`: part_len for modes 0/2, 1 for mode 1 */
    int32_t label_len;              /* pseudo labels per part in `abn_labels` (= --part_len) */
    int32_t l1_skip;                /* leading entries of the GLOBAL flat score vector excluded from the l1 mean */
    float lambda_1, lambda_MIL, lambda_aux, lambda_normal, lambda_abnormal;
    const float* out;               /* head output of this rank: [2*bs_local*part_num*score_len, c], c = 2 (mode 1) or 1;
                                       normal videos first, abnormal second (the reference's cat order) */
    const float* abn_labels;        /* [bs_local, part_num*label_len] pseudo labels of this rank's abnormal videos;
                                       NULL = no CE/BCE term (--temporal_only / mode 0) */
    const float* targets;           /* optional explicit soft targets [parts_local, 2] (parts = rows for mode 1,
                                       [2*bs_local*part_num] for mode 2); overrides abn_labels — used by the
                                       reference-named get_CE_loss / get_BCE_loss wrappers */
    float* bag;                     /* [2*bs_global] bag maxima, normal half then abnormal half.  phase 0 writes this
                                       rank's entries (caller zeroes first, then sum-all-reduces); phase 1 reads all */
    float* dout;                    /* same shape as `out`: d(rank's loss contribution)/d(out) */
    float* scalars;                 /* [5] loss, mil, err, l1, aux — this rank's contributions */
    int32_t phase;                  /* 0 = bags only; 1 = loss + gradient from `bag`; 2 = both (single rank) */
} LstcLossDesc;

int lstc_vad_loss(const LstcLossDesc* d, void* stream);

/* ------------------------------------------------------------------------ optimizer
 * torch.optim.Adagrad step as configured at Train/temporal_transformer_shanghaitech.py:83-85,142
 * (lr_decay=0, eps=1e-10, initial accumulator 0): g = grad*gscale + wd*w; s += g*g; w -= lr*g/(sqrt(s)+eps).
 * `gscale` carries the clip_grad_norm_ coefficient (:139-141) or 1. */
int lstc_adagrad_step(float* w, const float* grad, float* state, int64_t n, float lr, float weight_decay,
                      float eps, float gscale, void* stream);

/* optimizer.step() (Train/temporal_transformer_shanghaitech.py:142; Train/spatio_transformer_shanghaitech.py:109): the same
 * update for EVERY parameter of the step in ONE launch (torch.optim.Adagrad loops over the ~45 tensors; here the items ride in
 * the kernel arguments, 48 per launch).  `items` is a HOST array; element
 * arithmetic and order are those of lstc_adagrad_step - bit-identical results. */
typedef struct LstcAdagradItem {
    float* w;
    const float* grad;
    float* state;
    int64_t n;
    float lr, weight_decay, eps, grad_scale;
} LstcAdagradItem;
int lstc_adagrad_multi(const LstcAdagradItem* items, int32_t count, void* stream);
/* out[0] += sum(x^2) (f32 atomics over workgroup partials; caller zeroes) — for clip_grad_norm_. */
int lstc_sqnorm_accum(const float* x, int64_t n, float* out, void* stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) (Train/temporal_transformer_shanghaitech.py:139-141) over a whole
 * parameter list without a host read-back, so a captured step (HIP graph) can contain it:
 *   lstc_sqnorm_multi      out[0] = sum over every tensor of sum(x^2), out[1] = its square root (the total norm the reference
 *                          logs): one launch over all tensors (48 per launch) writing one
 *                          partial per 8192-element slice into `scratch` (>= lstc_sqnorm_multi_scratch(items, count) floats),
 *                          then one workgroup adds the partials in index order - bit-reproducible, no float atomics;
 *   lstc_clip_scale_multi  x *= max_norm / (sqrt(sqnorm[0]) + 1e-6) for every tensor when that coefficient is < 1 (torch's
 *                          clamp to 1; a NaN norm multiplies every tensor by NaN, as torch does), the coefficient formed ON THE
 *                          DEVICE from the value lstc_sqnorm_multi left there.
 * `items` is a HOST array (it rides in the kernel arguments). */
typedef struct LstcVecItem {
    float* x;
    int64_t n;
} LstcVecItem;
int64_t lstc_sqnorm_multi_scratch(const LstcVecItem* items, int32_t count);
int lstc_sqnorm_multi(const LstcVecItem* items, int32_t count, float* scratch, int64_t scratch_floats, float* out, void* stream);
int lstc_clip_scale_multi(const LstcVecItem* items, int32_t count, const float* sqnorm, float max_norm, void* stream);

/* x *= alpha in place: applies the clip_grad_norm_ coefficient to a gradient tensor. */
int lstc_scale(float* x, int64_t n, float alpha, void* stream);

/* f32 -> bf16 (round to nearest even) and back, n elements: the optional half-width gradient all-reduce of the data-parallel
 * path (replaces nothing upstream - nn.DataParallel reduces fp32, Train/temporal_transformer_shanghaitech.py:76-78 - and is
 * off by default: lstc_vad_amd/dist.py, reduce_dtype). */
int lstc_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);
int lstc_cast_bf16_f32(const void* x, float* y, int64_t n, void* stream);

/* ------------------------------------------------------------------------- data feed
 * dst[r, :] = src[idx[r], :] for r < n_rows, rows of `row_floats` contiguous floats (multiple of 4, 16-byte aligned
 * bases).  Forms the [B, part_num*part_len, n_patch, d] training batch out of an HBM-resident feature bank from the
 * clip indices the reference's sampler picks on the host (`feat[chosen[...], :]`, utils/load_dataset.py:88 +
 * default collate), so no feature bytes cross PCIe per step.  idx entries must lie in [0, src_rows). */
int lstc_gather_rows(const float* src, int64_t src_rows, const int64_t* idx, float* dst, int64_t n_rows,
                     int64_t row_floats, void* stream);

/* ------------------------------------------------------------------------------ misc */
int lstc_version(void);
const char* lstc_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* LSTC_HIP_H */
