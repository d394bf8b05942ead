// Shared by the attention translation units (csrc/attention.hip: first / second generation kernels, CLS kernels, the C-ABI entry
// points; csrc/attention_pk.hip: the packed-operand kernels of the bf16 mode).
#pragma once
#include "lstc_common.h"

namespace lstc_attn {

struct AttnParams {
    const float *Q, *K, *V;
    float* O;
    float* probs;
    const float* table;
    const int64_t* index;
    const float* dO;
    float *dQ, *dK, *dV;
    float* dtable;
    int N, S, H, dk, dv, ldq, ldk, ldv, ldo, index_ld, table_rows;
    float scale;
    DropKey dkey;
    int has_drop;
    int vec_qk, vec_v;     // float4 operand loads allowed for the dk / dv contractions
    int n_per_wg;
    int table_partials;     // 1: dtable is [gridDim.x][table_rows][H] partials (plain stores), 0: [rows][H] with atomics
    // bf16 mode (staged backward only): dQ / dK / dV written as packed bf16 operands [N*S, H*dk|dv] (lstc_pack1 layout) instead
    // of f32 - they are consumed only by the packed weight-gradient and input-gradient GEMMs
    void *dQp, *dKp, *dVp;
    int kbq, kbk, kbv;      // 32-k tiles per 128-row block of each pack
    int tq0, tk0, tv0;      // first k tile of the dQ / dK / dV columns inside their pack (one fused pack: column offsets / 32)
    void* Op;               // forward, bf16 mode: O written as a packed bf16 operand [N*S, H*dv] instead of f32
    int kbo;
    // bf16 mode, packed INPUTS (third-generation kernels): Q / K / V / dO are lstc_pack1 buffers; 32-k tiles per 128-row block of
    // each pack and the first tile of head 0's columns inside it
    const __bf16 *Qi, *Ki, *Vi, *dOi;
    int kiq, kik, kiv, kido;
    int iq0, ik0, iv0, ido0;
    int pld;                // row pitch of `probs` in floats (packed-input kernels: a multiple of 4, >= S; else S)
};

typedef __bf16 attn_h8 __attribute__((ext_vector_type(8)));
typedef float attn_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 attn_h2 __attribute__((ext_vector_type(2)));

// csrc/attention_pk.hip: launch the packed-operand kernels (LstcAttnDesc.in_pack_cols > 0) for T = ceil(S / 32) in 1..3 over
// `chunks` groups of p.n_per_wg sequences x p.H heads; 0 or LSTC_E_* / hipError_t as the entry points return it
// (internal to the library: not part of the C ABI, hidden from the dynamic symbol table)
__attribute__((visibility("hidden"))) int attn3_fwd_launch(const AttnParams& p, int T, int chunks, hipStream_t st);
__attribute__((visibility("hidden"))) int attn3_bwd_launch(const AttnParams& p, int T, int chunks, hipStream_t st);
// exact-f32 forward on the same structure (f32 operands, dense probs): csrc/attention_pk.hip
__attribute__((visibility("hidden"))) int attn3f_fwd_launch(const AttnParams& p, int T, int chunks, hipStream_t st);

}  // namespace lstc_attn
