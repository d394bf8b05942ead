// bf16-compute GEMM with f32 storage (LstcGemmDesc.dtype = LSTC_BF16) for gfx950.
//
// Same contract and epilogues as gemm_f32.hip — A, B, C, bias, residual stay float32 in HBM — but the operands are
// rounded to bf16 (RNE) while they are staged into LDS and the contraction runs on v_mfma_f32_32x32x16_bf16 with f32
// accumulation.  This is the "bf16" mode of BASELINE.json configs 3 / 5: fp32 master weights and activations, bf16
// matrix cores (16x the f32 MFMA rate).  Because the operands still arrive as f32 (2x the bytes of a bf16 pipeline)
// the kernel is bound by L2->LDS operand traffic, not by the MFMA pipe: 128x128x64 tiles need 64 KB per 2.1 MFLOP.
//
// Layout notes (cdna_hip_programming 3 / T10):
//  * K-contiguous operands: LDS image [rows][64 bf16 + 8 pad] = 144-B rows — the same conflict-free geometry as the
//    f32 kernel; lane-half h owns k = 32h..32h+31 of the 64-deep tile and reads one ds_read_b128 (8 bf16) per MFMA.
//  * M/N-contiguous (k-major) operands: LDS image [64 k][rows + 32 pad] bf16 (320-B rows for 128 columns); the MFMA
//    fragment (8 consecutive k for one column) is fetched with two ds_read_b64_tr_b16 (hardware transpose read:
//    a 16-lane group reads a 4-row x 16-column block and each lane receives one column), conflict-free because
//    consecutive k rows land 64 B apart modulo the 256-B bank row.
#include "lstc_common.h"

namespace {

constexpr int BK = 64;
constexpr int KC_LD = 72;          // bf16 per row of a K-contiguous image (64 + 8 pad = 144 B)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

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
};

template <int R>
constexpr int mc_ld() { return R + 32; }      // bf16 per k row of a k-major image

template <int R, bool KC>
constexpr int stage_elems() { return KC ? R * KC_LD : BK * mc_ld<R>(); }

__device__ __forceinline__ bf16x4 to_bf16(const float4& v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    return o;
}

template <int R, int NT, bool KC, bool VEC>
struct Stager {
    static constexpr int NV = R * 16 / NT;      // float4 per thread per K tile
    float4 v[NV];

    template <bool CHECK_K>
    __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int r0, int r_total, int k0, int K) {
        const int t = threadIdx.x;
        if (KC) {
            const int k = k0 + (t & 15) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = min(r0 + (t >> 4) + i * (NT / 16), r_total - 1);
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
            constexpr int CPR = R / 4;
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

    __device__ __forceinline__ void store(__bf16* __restrict__ lds) const {
        const int t = threadIdx.x;
        if (KC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = (t >> 4) + i * (NT / 16);
                *reinterpret_cast<bf16x4*>(lds + r * KC_LD + (t & 15) * 4) = to_bf16(v[i]);
            }
        } else {
            constexpr int CPR = R / 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = t / CPR + i * (NT / CPR);
                *reinterpret_cast<bf16x4*>(lds + k * mc_ld<R>() + (t % CPR) * 4) = to_bf16(v[i]);
            }
        }
    }
};

// Fragment of MFMA k-step s (0..3) for the 32-row/column operand tile starting at `tile0`:
// lane (r = lane&31, h = lane>>5) gets k = 32h + 8s .. +7 of row/column tile0 + r.
template <int R, bool KC>
__device__ __forceinline__ bf16x8 read_frag(const __bf16* __restrict__ lds, int tile0, int s) {
    const int lane = threadIdx.x & 63;
    if (KC) {
        const int r = lane & 31, h = lane >> 5;
        return *reinterpret_cast<const bf16x8*>(lds + (tile0 + r) * KC_LD + 32 * h + 8 * s);
    } else {
        // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block and
        // receives column (lane&15) of the 4 rows.  Groups 0/1 = columns 0-15 / 16-31 for h = 0, groups 2/3 for h = 1.
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int h = g >> 1, col = tile0 + 16 * (g & 1) + 4 * pp;
        const int k0 = 32 * h + 8 * s;
        const __bf16* a0 = lds + (k0 + q) * mc_ld<R>() + col;
        const __bf16* a1 = a0 + 4 * mc_ld<R>();
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a1));
        typedef short short8v __attribute__((ext_vector_type(8)));
        short8v f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return __builtin_bit_cast(bf16x8, f);
    }
}

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, bool VA, bool VB>
__global__ void __launch_bounds__(WGM* WGN * 64) gemm_bf16c_kernel(const GemmParams p_in) {
    GemmParams p = p_in;
    p.dk = drop_key_now(p.dk);          // graph replays: seed + device offset (lstc_dropout_seed_device)
    p.A += (size_t)blockIdx.z * p.batch_stride_a;
    p.B += (size_t)blockIdx.z * p.batch_stride_b;
    p.C += (size_t)blockIdx.z * p.batch_stride_c;
    constexpr int NT = WGM * WGN * 64;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_ST = stage_elems<BM, A_KC>();
    constexpr int B_ST = stage_elems<BN, B_KC>();
    extern __shared__ __attribute__((aligned(16))) __bf16 smem_bf[];
    __bf16* const As = smem_bf;
    __bf16* const Bs = smem_bf + 2 * A_ST;

    const int nwg = p.tilesM * p.tilesN;
    int pid = blockIdx.x;
    {
        const int xcd = pid & 7, idx = pid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = pid / p.tilesN, nt = pid % p.tilesN;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt0 = blockIdx.y * p.ktiles_per_split;
    const int kt1 = min(p.ktiles, kt0 + p.ktiles_per_split);
    const int nkt = kt1 - kt0;

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
    const bool k_tail = (p.K % BK) != 0;
    auto gload = [&](int kt) {
        if (k_tail && kt == p.ktiles - 1) {
            sa.template load<true>(p.A, p.lda, m0, p.M, kt * BK, p.K);
            sb.template load<true>(p.B, p.ldb, n0, p.N, kt * BK, p.K);
        } else {
            sa.template load<false>(p.A, p.lda, m0, p.M, kt * BK, p.K);
            sb.template load<false>(p.B, p.ldb, n0, p.N, kt * BK, p.K);
        }
    };

    if (nkt > 0) {
        gload(kt0);
        sa.store(As);
        sb.store(Bs);
    }
    __syncthreads();
    for (int it = 0; it < nkt; ++it) {
        const int cur = it & 1;
        const bool more = it + 1 < nkt;
        if (more) gload(kt0 + it + 1);
        const __bf16* a_lds = As + cur * A_ST;
        const __bf16* b_lds = Bs + cur * B_ST;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, A_KC>(a_lds, wm * WTM + i * 32, s);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = read_frag<BN, B_KC>(b_lds, wn * WTN + j * 32, s);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            sa.store(As + (cur ^ 1) * A_ST);
            sb.store(Bs + (cur ^ 1) * B_ST);
        }
        __syncthreads();
    }

    // ---- epilogue (identical semantics to gemm_f32.hip)
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

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC>
int launch_cfg(const GemmParams& p, bool va, bool vb, int splits, hipStream_t st) {
    constexpr int NT = WGM * WGN * 64;
    constexpr size_t lds = (size_t)(2 * stage_elems<BM, A_KC>() + 2 * stage_elems<BN, B_KC>()) * sizeof(__bf16);
    dim3 grid(p.tilesM * p.tilesN, splits, p.batch), block(NT);
#define LSTC_GO(VA, VB)                                                                                          \
    do {                                                                                                         \
        auto kern = gemm_bf16c_kernel<BM, BN, WGM, WGN, A_KC, B_KC, VA, VB>;                                     \
        static LstcDevOnce attr_done;                                                                            \
        const int dev_ = attr_done.begin();                                                                      \
        if (dev_ >= 0) {                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                 \
            attr_done.end(dev_);                                                                                 \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, grid, block, lds, st, p);                                                       \
    } while (0)
    if (va && vb) LSTC_GO(true, true);
    else if (va) LSTC_GO(true, false);
    else if (vb) LSTC_GO(false, true);
    else LSTC_GO(false, false);
#undef LSTC_GO
    return lstc_launch_status();
}

template <bool A_KC, bool B_KC>
int launch_layout(GemmParams& p, bool va, bool vb, int splits, int variant, hipStream_t st) {
    // variant 0 = default: 256x128 (8 waves) except NT with deep K, where 128x128 measured faster (MI355X, LTN shapes,
    // TFLOP/s 128x128 vs 256x128: NT K=2048 387/407, NT K=4096 395/374, NN 416/463, TN split-4 473/531); 1 = 128x128; 2 = 256x128
    if (variant == 0) variant = (A_KC && B_KC && p.K >= 4096) ? 1 : 2;
    const int BM = variant == 2 ? 256 : 128, BN = 128;
    p.tilesM = (p.M + BM - 1) / BM;
    p.tilesN = (p.N + BN - 1) / BN;
    if (variant == 2) return launch_cfg<256, 128, 4, 2, A_KC, B_KC>(p, va, vb, splits, st);
    return launch_cfg<128, 128, 2, 2, A_KC, B_KC>(p, va, vb, splits, st);
}

}  // namespace

__attribute__((visibility("hidden"))) int lstc_gemm_bf16_impl(const LstcGemmDesc* d, hipStream_t st) {
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
    const int eff_splits = (p.ktiles + p.ktiles_per_split - 1) / p.ktiles_per_split;
    const bool va = aligned16(d->A) && (d->lda % 4 == 0) && ((d->transA ? d->M : d->K) % 4 == 0) &&
                    (p.batch <= 1 || d->batch_stride_a % 4 == 0);
    const bool vb = aligned16(d->B) && (d->ldb % 4 == 0) && ((d->transB ? d->K : d->N) % 4 == 0) &&
                    (p.batch <= 1 || d->batch_stride_b % 4 == 0);
    if (!d->transA && d->transB) return launch_layout<true, true>(p, va, vb, eff_splits, d->variant & 15, st);
    if (!d->transA && !d->transB) return launch_layout<true, false>(p, va, vb, eff_splits, d->variant & 15, st);
    return launch_layout<false, false>(p, va, vb, eff_splits, d->variant & 15, st);
}
