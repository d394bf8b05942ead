// HBM-bound row-wise kernels of the training step (gfx950): LayerNorm fwd/bwd, CLS-mean + concat,
// column sums, dropout replay, the fused last head layer, Adagrad, squared-norm.
// All of them stream their operands once with 16-B-per-lane accesses where the layout allows
// (cdna_hip_programming Guideline 13) and reduce with wave shuffles.
#include "lstc_common.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------- LayerNorm
// One wave per row; the row lives in registers (NV float4 per lane, d <= 256*NV) so x is read once.
// Two-pass statistics (mean, then centred variance) like ATen's CPU kernel for parity.
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

template <int NV>
__global__ void __launch_bounds__(NT) ln_fwd_vec(const float* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ y,
                                                  float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                  int d, float eps, __bf16* __restrict__ packed, int KBp) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NT / 64);
    const int nv4 = d >> 2;
    float4 g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv4) {
            g[i] = reinterpret_cast<const float4*>(gamma)[c];
            b[i] = reinterpret_cast<const float4*>(beta)[c];
        }
    }
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float4* xr = reinterpret_cast<const float4*>(x + r * d);
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < nv4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv4) {
                const float a0 = v[i].x - mu, a1 = v[i].y - mu, a2 = v[i].z - mu, a3 = v[i].w - mu;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rs = 1.f / sqrtf(wave_sum(q) / (float)d + eps);
        float4* yr = reinterpret_cast<float4*>(y + r * d);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv4) {
                float4 o;
                o.x = (v[i].x - mu) * rs * g[i].x + b[i].x;
                o.y = (v[i].y - mu) * rs * g[i].y + b[i].y;
                o.z = (v[i].z - mu) * rs * g[i].z + b[i].z;
                o.w = (v[i].w - mu) * rs * g[i].w + b[i].w;
                yr[c] = o;
                if (packed) {       // the same row in the packed bf16 layout of the next GEMM's A operand: 8 B per lane,
                    bf16x4v h;      // 8 lanes = one tile row's 64 contiguous bytes
                    h[0] = (__bf16)o.x; h[1] = (__bf16)o.y; h[2] = (__bf16)o.z; h[3] = (__bf16)o.w;
                    *reinterpret_cast<bf16x4v*>(packed + p1_offset(r, 4 * c, KBp)) = h;
                }
            }
        }
        if (lane == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
}

// Any d (odd widths, > 2048): scalar, three passes over the row (re-reads hit L1/L2).
__global__ void __launch_bounds__(NT) ln_fwd_generic(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                      int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NT / 64);
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float* xr = x + r * d;
        float s = 0.f;
        for (int c = lane; c < d; c += 64) s += xr[c];
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float a = xr[c] - mu;
            q += a * a;
        }
        const float rs = 1.f / sqrtf(wave_sum(q) / (float)d + eps);
        for (int c = lane; c < d; c += 64) y[r * d + c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
        if (lane == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
}

// dx = rstd * (g*dy - mean_d(g*dy) - xhat * mean_d(g*dy*xhat)); per-workgroup partial dgamma = sum dy*xhat,
// dbeta = sum dy written to partial[0][blockIdx.x][:] / partial[1][blockIdx.x][:].
// PACK: the kernel also emits what the two GEMMs behind the residual branch consume - df = dropout-replay(dx) rounded to
// bf16 in the lstc_pack1 layout (dropout index = flat row * d + col, as lstc_dropout_apply) - and a third partial, the
// column sums of df (the bias gradient of the Linear in front of the dropout).
template <int NV, bool PACK>
__global__ void __launch_bounds__(NT, NV == 8 ? 2 : 1) ln_bwd_vec(const float* __restrict__ dy, const float* __restrict__ x,
                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, float* __restrict__ dx,
                                                  float* __restrict__ partial, int64_t rows, int d,
                                                  __bf16* __restrict__ packed, int KBp, DropKey key) {
    key = drop_key_now(key);
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [NT/64][d], reused per partial kind
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * (NT / 64) + wv;
    const int64_t nwaves = (int64_t)gridDim.x * (NT / 64);
    const int nv4 = d >> 2;
    // PACK carries a third accumulator row: gamma is then re-read per row (8 KB, L1-resident) instead of held in registers
    float4 g[PACK ? 1 : NV], ag[NV], ab[NV], ad[PACK ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (!PACK) g[i] = c < nv4 ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PACK) ad[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float4* xr = reinterpret_cast<const float4*>(x + r * d);
        const float4* dr = reinterpret_cast<const float4*>(dy + r * d);
        const float mu = mean[r], rs = rstd[r];
        float4 xh[NV], gd[NV];
        // per-lane sums in two halves (i < NVa, i >= NVa), added before the butterfly: the order ln_bwd_pack2's wave pair
        // reproduces, so both kernels give the same dx bit for bit
        constexpr int NVa = (NV + 1) / 2;
        float s1 = 0.f, s2 = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv4) {
                const float4 xv = xr[c], dv = dr[c];
                const float4 gi = PACK ? reinterpret_cast<const float4*>(gamma)[c] : g[PACK ? 0 : i];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                gd[i] = make_float4(dv.x * gi.x, dv.y * gi.y, dv.z * gi.z, dv.w * gi.w);
                const float t1 = (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
                const float t2 = (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
                if (i < NVa) { s1 += t1; s2 += t2; } else { s1b += t1; s2b += t2; }
                ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
            } else {
                xh[i] = gd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float m1 = wave_sum(s1 + s1b) / (float)d, m2 = wave_sum(s2 + s2b) / (float)d;
        float4* o = reinterpret_cast<float4*>(dx + r * d);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv4) {
                const float4 ov = make_float4(rs * (gd[i].x - m1 - xh[i].x * m2), rs * (gd[i].y - m1 - xh[i].y * m2),
                                              rs * (gd[i].z - m1 - xh[i].z * m2), rs * (gd[i].w - m1 - xh[i].w * m2));
                o[c] = ov;
                if (PACK) {
                    const uint32_t fi = (uint32_t)(r * d) + 4u * (uint32_t)c;
                    float4 f;
                    f.x = drop_keep(fi + 0, key) ? ov.x * key.scale : 0.f;
                    f.y = drop_keep(fi + 1, key) ? ov.y * key.scale : 0.f;
                    f.z = drop_keep(fi + 2, key) ? ov.z * key.scale : 0.f;
                    f.w = drop_keep(fi + 3, key) ? ov.w * key.scale : 0.f;
                    ad[i].x += f.x; ad[i].y += f.y; ad[i].z += f.z; ad[i].w += f.w;
                    bf16x4v h;
                    h[0] = (__bf16)f.x; h[1] = (__bf16)f.y; h[2] = (__bf16)f.z; h[3] = (__bf16)f.w;
                    *reinterpret_cast<bf16x4v*>(packed + p1_offset(r, 4 * c, KBp)) = h;
                }
            }
        }
    }
    // combine the workgroup's 4 waves through LDS, one partial row per workgroup and kind
    float4* s4 = reinterpret_cast<float4*>(sm);
#pragma unroll
    for (int which = 0; which < (PACK ? 3 : 2); ++which) {
        if (which) __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv4) s4[wv * nv4 + c] = which == 0 ? ag[i] : which == 1 ? ab[i] : ad[PACK ? i : 0];
        }
        __syncthreads();
        for (int cc = threadIdx.x; cc < nv4; cc += NT) {
            float4 t = s4[cc];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) {
                const float4 u = s4[w * nv4 + cc];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            reinterpret_cast<float4*>(partial + ((size_t)which * gridDim.x + blockIdx.x) * d)[cc] = t;
        }
    }
}

// Two waves per row (each NVH float4 per lane of its half of the columns): half the live registers of ln_bwd_vec<.., true>,
// so three accumulator rows (dgamma, dbeta, column sums of df) fit at 3 waves/SIMD.  The row statistics cross the wave pair
// through LDS (double-buffered by iteration parity: one barrier per row pair).
// MODE 0: dx and the two partials only; 1: + packed bf16 df + third partial (bf16 mode); 2: + f32 df + third partial (f32 modes:
// `packed` is then a float [rows, d] - the dropout replay needs no pass of its own in any mode).
template <int NVH, int MODE>
__global__ void __launch_bounds__(NT) ln_bwd_pack2(const float* __restrict__ dy, const float* __restrict__ x,
                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, float* __restrict__ dx,
                                                    float* __restrict__ partial, int64_t rows, int d,
                                                    __bf16* __restrict__ packed, int KBp, DropKey key) {
    key = drop_key_now(key);
    constexpr bool PACK = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [2][d] combine buffer, then [2][4][64][2] per-lane row sums
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, slot = wv >> 1, half = wv & 1;
    const int nv4 = d >> 2, hv4 = nv4 >> 1;
    float* red = sm + 2 * d;
    float4 g[NVH], ag[NVH], ab[NVH], ad[PACK ? NVH : 1];
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
        const int cl = lane + 64 * i;
        g[i] = cl < hv4 ? reinterpret_cast<const float4*>(gamma)[half * hv4 + cl] : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PACK) ad[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int64_t stride = (int64_t)gridDim.x * 2;
    const int64_t iters = (rows + stride - 1) / stride;
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t r = it * stride + (int64_t)blockIdx.x * 2 + slot;
        const bool live = r < rows;
        const int64_t rr = live ? r : 0;
        const float4* xr = reinterpret_cast<const float4*>(x + rr * d) + half * hv4;
        const float4* dr = reinterpret_cast<const float4*>(dy + rr * d) + half * hv4;
        const float mu = mean[rr], rs = rstd[rr];
        float4 xh[NVH], gd[NVH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int cl = lane + 64 * i;
            if (cl < hv4 && live) {
                const float4 xv = xr[cl], dv = dr[cl];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                gd[i] = make_float4(dv.x * g[i].x, dv.y * g[i].y, dv.z * g[i].z, dv.w * g[i].w);
                s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
                s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
                ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
            } else {
                xh[i] = gd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float2* rb = reinterpret_cast<float2*>(red) + (it & 1) * 256;
        rb[wv * 64 + lane] = make_float2(s1, s2);
        __syncthreads();
        // lane l of both waves adds the pair's per-lane halves in the same order (columns below d/2 first), then the same
        // butterfly: the pair agrees bit for bit, and with ln_bwd_vec's two-half order
        const float2 pa = rb[(slot * 2) * 64 + lane], pb = rb[(slot * 2 + 1) * 64 + lane];
        const float m1 = wave_sum(pa.x + pb.x) / (float)d, m2 = wave_sum(pa.y + pb.y) / (float)d;
        if (!live) continue;
        float4* o = reinterpret_cast<float4*>(dx + r * d) + half * hv4;
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int cl = lane + 64 * i;
            if (cl < hv4) {
                const float4 ov = make_float4(rs * (gd[i].x - m1 - xh[i].x * m2), rs * (gd[i].y - m1 - xh[i].y * m2),
                                              rs * (gd[i].z - m1 - xh[i].z * m2), rs * (gd[i].w - m1 - xh[i].w * m2));
                o[cl] = ov;
                if (!PACK) continue;
                const int c = half * hv4 + cl;
                const uint32_t fi = (uint32_t)(r * d) + 4u * (uint32_t)c;
                float4 f;
                f.x = drop_keep(fi + 0, key) ? ov.x * key.scale : 0.f;
                f.y = drop_keep(fi + 1, key) ? ov.y * key.scale : 0.f;
                f.z = drop_keep(fi + 2, key) ? ov.z * key.scale : 0.f;
                f.w = drop_keep(fi + 3, key) ? ov.w * key.scale : 0.f;
                ad[i].x += f.x; ad[i].y += f.y; ad[i].z += f.z; ad[i].w += f.w;
                if constexpr (MODE == 2) {
                    reinterpret_cast<float4*>(reinterpret_cast<float*>(packed) + r * d)[c] = f;
                } else {
                    bf16x4v h;
                    h[0] = (__bf16)f.x; h[1] = (__bf16)f.y; h[2] = (__bf16)f.z; h[3] = (__bf16)f.w;
                    *reinterpret_cast<bf16x4v*>(packed + p1_offset(r, 4 * c, KBp)) = h;
                }
            }
        }
    }
    // combine the two row slots through LDS, one partial row per workgroup and kind
    float4* s4 = reinterpret_cast<float4*>(sm);
#pragma unroll
    for (int which = 0; which < (PACK ? 3 : 2); ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int cl = lane + 64 * i;
            if (cl < hv4) s4[slot * nv4 + half * hv4 + cl] = which == 0 ? ag[i] : which == 1 ? ab[i] : ad[PACK ? i : 0];
        }
        __syncthreads();
        for (int cc = threadIdx.x; cc < nv4; cc += NT) {
            const float4 t = s4[cc], u = s4[nv4 + cc];
            reinterpret_cast<float4*>(partial + ((size_t)which * gridDim.x + blockIdx.x) * d)[cc] =
                make_float4(t.x + u.x, t.y + u.y, t.z + u.z, t.w + u.w);
        }
    }
}

// ---- bf16 activation stream (round 5; --dtype bf16 with bf16 activations).  Between the CLS concat and the last full layer's
// LayerNorm no f32 activation exists: the residual sums leave the GEMM epilogues as lstc_pack1 operands (LSTC_EPI_OUT_PACK +
// LSTC_EPI_RESIDUAL_PACK), the LayerNorm reads that pack and writes the pack the next block consumes (A operand AND residual),
// and the gradient of the residual stream travels the same way.  A lane owns 16-B chunks (8 consecutive columns) of its row:
// chunk c8 of row r is one 16-B access at p1_offset(r, 8 c8, KBp); 4 lanes cover a tile row's 64 contiguous bytes.
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

// One wave per row, NC = d / 512 chunks per lane.  XP: x is a pack (else f32 [rows, d]); YF / YP: write y as f32 / as a pack.
// (Measured and not kept: the next row's chunks prefetched before the current row is reduced - with gamma / beta in registers
// that is 174 VGPRs = 2 waves per SIMD; with gamma / beta in LDS 124 VGPRs = 4 waves, and 221 us instead of 173 us per 100352 x
// 2048 launch: the reduction waits on the LDS reads.  The plain form below is the fastest of the three.)
template <int NC, bool XP, bool YF, bool YP>
__global__ void __launch_bounds__(NT) ln_fwd_act(const float* __restrict__ x, const __bf16* __restrict__ xp,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ y, __bf16* __restrict__ yp, float* __restrict__ mean,
                                                  float* __restrict__ rstd, int64_t rows, int d, float eps, int KBp) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NT / 64);
    float g[NC][8], b[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c0 = 8 * (lane + 64 * i);
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
        g[i][0] = g0.x; g[i][1] = g0.y; g[i][2] = g0.z; g[i][3] = g0.w; g[i][4] = g1.x; g[i][5] = g1.y; g[i][6] = g1.z; g[i][7] = g1.w;
        b[i][0] = b0.x; b[i][1] = b0.y; b[i][2] = b0.z; b[i][3] = b0.w; b[i][4] = b1.x; b[i][5] = b1.y; b[i][6] = b1.z; b[i][7] = b1.w;
    }
    for (int64_t r = wave0; r < rows; r += nwaves) {
        float v[NC][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c0 = 8 * (lane + 64 * i);
            if constexpr (XP) {
                const bf16x8v h = *reinterpret_cast<const bf16x8v*>(xp + p1_offset(r, c0, KBp));
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = (float)h[j];
            } else {
                const float4 a0 = *reinterpret_cast<const float4*>(x + r * d + c0), a1 = *reinterpret_cast<const float4*>(x + r * d + c0 + 4);
                v[i][0] = a0.x; v[i][1] = a0.y; v[i][2] = a0.z; v[i][3] = a0.w; v[i][4] = a1.x; v[i][5] = a1.y; v[i][6] = a1.z; v[i][7] = a1.w;
            }
            s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
        }
        const float mu = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = v[i][j] - mu; q += a * a; }
        const float rs = 1.f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c0 = 8 * (lane + 64 * i);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mu) * rs * g[i][j] + b[i][j];
            if constexpr (YF) {
                *reinterpret_cast<float4*>(y + r * d + c0) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(y + r * d + c0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
            if constexpr (YP) {
                bf16x8v h;
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = (__bf16)o[j];
                *reinterpret_cast<bf16x8v*>(yp + p1_offset(r, c0, KBp)) = h;
            }
        }
        if (lane == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
}

// Backward of z = LayerNorm(dropout(f) + x) on packs: two waves per row (NCH = d / 1024 chunks per lane each), the structure of
// ln_bwd_pack2.  xp = pack of the pre-LayerNorm sum; the incoming gradient is a pack (DYP) or f32; DX: the gradient of the
// residual sum leaves as a pack (the residual operand of the block's input-gradient GEMM) - not at all when nobody reads it
// (layer 0).  dfp = pack of dropout-replay(dx); partial [3, gridDim.x, d] = dgamma, dbeta, column sums of df.
template <int NCH, bool DYP, bool DX>
__global__ void __launch_bounds__(NT) ln_bwd_act(const float* __restrict__ dy, const __bf16* __restrict__ dyp,
                                                  const __bf16* __restrict__ xp, const float* __restrict__ gamma,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                  __bf16* __restrict__ dxp, float* __restrict__ partial, int64_t rows, int d,
                                                  __bf16* __restrict__ dfp, int KBp, DropKey key) {
    key = drop_key_now(key);
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [2][d] combine buffer, then [2][4][64][2] per-lane row sums
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, slot = wv >> 1, half = wv & 1;
    const int hc = d >> 4;                                        // 16-B chunks per half row
    float* red = sm + 2 * d;
    float g[NCH][8], ag[NCH][8], ab[NCH][8], ad[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = 8 * (half * hc + lane + 64 * i);
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
        g[i][0] = g0.x; g[i][1] = g0.y; g[i][2] = g0.z; g[i][3] = g0.w; g[i][4] = g1.x; g[i][5] = g1.y; g[i][6] = g1.z; g[i][7] = g1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) ag[i][j] = ab[i][j] = ad[i][j] = 0.f;
    }
    const int64_t stride = (int64_t)gridDim.x * 2;
    const int64_t iters = (rows + stride - 1) / stride;
    // operands of iteration it + 1 are requested before iteration it is reduced (raw bf16: 8 - 16 registers per tensor): twice the
    // bytes in flight per wave.  Rows beyond the end are clamped to row 0 (never used: ``live``).
    bf16x8v xc[NCH], xn[NCH], dc[DYP ? NCH : 1], dn[DYP ? NCH : 1];
    float4 fc[DYP ? 1 : 2 * NCH], fn[DYP ? 1 : 2 * NCH];
    auto fetch = [&](int64_t it, bf16x8v* xo, bf16x8v* dpo, float4* dfo) {
        const int64_t r = it * stride + (int64_t)blockIdx.x * 2 + slot;
        const int64_t rr = r < rows ? r : 0;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c0 = 8 * (half * hc + lane + 64 * i);
            const size_t off = p1_offset(rr, c0, KBp);
            xo[i] = *reinterpret_cast<const bf16x8v*>(xp + off);
            if constexpr (DYP) dpo[i] = *reinterpret_cast<const bf16x8v*>(dyp + off);
            else {
                dfo[2 * i] = *reinterpret_cast<const float4*>(dy + rr * d + c0);
                dfo[2 * i + 1] = *reinterpret_cast<const float4*>(dy + rr * d + c0 + 4);
            }
        }
    };
    if (iters > 0) fetch(0, xc, dc, fc);
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t r = it * stride + (int64_t)blockIdx.x * 2 + slot;
        const bool live = r < rows;
        const int64_t rr = live ? r : 0;
        if (it + 1 < iters) fetch(it + 1, xn, dn, fn);
        const float mu = mean[rr], rs = rstd[rr];
        float xh[NCH][8], gd[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (live) {
                float dv[8];
                if constexpr (DYP) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) dv[j] = (float)dc[i][j];
                } else {
                    const float4 a0 = fc[2 * i], a1 = fc[2 * i + 1];
                    dv[0] = a0.x; dv[1] = a0.y; dv[2] = a0.z; dv[3] = a0.w; dv[4] = a1.x; dv[5] = a1.y; dv[6] = a1.z; dv[7] = a1.w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[i][j] = ((float)xc[i][j] - mu) * rs;
                    gd[i][j] = dv[j] * g[i][j];
                    s1 += gd[i][j];
                    s2 += gd[i][j] * xh[i][j];
                    ag[i][j] += dv[j] * xh[i][j];
                    ab[i][j] += dv[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xh[i][j] = gd[i][j] = 0.f;
            }
        }
        float2* rb = reinterpret_cast<float2*>(red) + (it & 1) * 256;
        rb[wv * 64 + lane] = make_float2(s1, s2);
        __syncthreads();
        const float2 pa = rb[(slot * 2) * 64 + lane], pb = rb[(slot * 2 + 1) * 64 + lane];
        const float m1 = wave_sum(pa.x + pb.x) / (float)d, m2 = wave_sum(pa.y + pb.y) / (float)d;
        if (live) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c0 = 8 * (half * hc + lane + 64 * i);
                const uint32_t fi = (uint32_t)(r * d) + (uint32_t)c0;
                bf16x8v ho, hf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float ov = rs * (gd[i][j] - m1 - xh[i][j] * m2);
                    const float f = drop_keep(fi + j, key) ? ov * key.scale : 0.f;
                    ad[i][j] += f;
                    ho[j] = (__bf16)ov;
                    hf[j] = (__bf16)f;
                }
                const size_t off = p1_offset(r, c0, KBp);
                if constexpr (DX) *reinterpret_cast<bf16x8v*>(dxp + off) = ho;
                *reinterpret_cast<bf16x8v*>(dfp + off) = hf;
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            xc[i] = xn[i];
            if constexpr (DYP) dc[i] = dn[i];
            else { fc[2 * i] = fn[2 * i]; fc[2 * i + 1] = fn[2 * i + 1]; }
        }
    }
    // combine the two row slots through LDS, one partial row per workgroup and kind
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c0 = 8 * (half * hc + lane + 64 * i);
            float* dst = sm + slot * d + c0;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = which == 0 ? ag[i][j] : which == 1 ? ab[i][j] : ad[i][j];
        }
        __syncthreads();
        for (int cc = threadIdx.x; cc < (d >> 2); cc += NT) {
            const float4 t = reinterpret_cast<const float4*>(sm)[cc], u = reinterpret_cast<const float4*>(sm)[(d >> 2) + cc];
            reinterpret_cast<float4*>(partial + ((size_t)which * gridDim.x + blockIdx.x) * d)[cc] =
                make_float4(t.x + u.x, t.y + u.y, t.z + u.z, t.w + u.w);
        }
    }
}

__global__ void __launch_bounds__(NT) ln_bwd_generic(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, float* __restrict__ dx,
                                                      float* __restrict__ partial, int64_t rows, int d) {
    // one workgroup = one partial row; each wave walks rows, threads accumulate columns in LDS-free fashion
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [2][d] accumulated with LDS atomics
    for (int c = threadIdx.x; c < 2 * d; c += NT) sm[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NT / 64);
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float mu = mean[r], rs = rstd[r];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float xh = (x[r * d + c] - mu) * rs, gd = dy[r * d + c] * gamma[c];
            s1 += gd;
            s2 += gd * xh;
        }
        const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
        for (int c = lane; c < d; c += 64) {
            const float xh = (x[r * d + c] - mu) * rs, dv = dy[r * d + c];
            dx[r * d + c] = rs * (dv * gamma[c] - m1 - xh * m2);
            atomicAdd(&sm[c], dv * xh);
            atomicAdd(&sm[d + c], dv);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * d; c += NT) {
        const int which = c / d, cc = c % d;
        partial[((size_t)which * gridDim.x + blockIdx.x) * d + cc] = sm[c];
    }
}

// ------------------------------------------------------------------------------ column sums of a packed bf16 operand
// partial[rg][k] = sum over the rows of row-block group rg of the lstc_pack1 operand [rows, K] (bf16 -> f32 adds): the bias
// gradient of a Linear whose output gradient exists only in packed form (LSTC_EPI_OUT_PACK).  One workgroup per 32-k tile and
// row-block group; a thread owns one 16-B chunk (8 k) of rows r and r + 64 of every tile of its group.
__global__ void __launch_bounds__(NT) colsum_pack1_kernel(const __bf16* __restrict__ pk, int RB, int KBp, int K,
                                                           float* __restrict__ partial) {
    __shared__ float red[64][33];
    const int kb = blockIdx.x, rg = blockIdx.y, ngrp = gridDim.y;
    const int per = (RB + ngrp - 1) / ngrp, rb0 = rg * per, rb1 = min(RB, rb0 + per);
    const int r = threadIdx.x >> 2, c = threadIdx.x & 3;             // row (and row + 64), LOGICAL chunk
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int rb = rb0; rb < rb1; ++rb) {
        const bf16x8v* t = reinterpret_cast<const bf16x8v*>(pk + ((size_t)rb * KBp + kb) * 4096);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int rr = r + 64 * hh;
            const bf16x8v v = t[rr * 4 + (c ^ ((rr >> 2) & 3))];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[r][c * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.f;
        for (int i = 0; i < 64; ++i) s += red[i][threadIdx.x];
        const int k = kb * 32 + threadIdx.x;
        if (k < K) partial[(size_t)rg * K + k] = s;
    }
}

// ------------------------------------------------------------------------------ CLS concat
// grid (N, ceil(d/NT)); thread owns one column (coalesced across the workgroup), walks the tokens.
__global__ void __launch_bounds__(NT) cls_concat_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x_hi,
                                                             int64_t n_lo, const float* __restrict__ cls,
                                                             const float* __restrict__ pos, float* __restrict__ y, int S,
                                                             int d, __bf16* __restrict__ packed, int KBp) {
    const int64_t n = blockIdx.x;
    const int c = blockIdx.y * NT + threadIdx.x;
    if (c >= d) return;
    // sequences [0, n_lo) come from x, [n_lo, N) from x_hi: the reference's cat([normal, abnormal]) fused away
    const float* xr = (x_hi && n >= n_lo ? x_hi + (n - n_lo) * (int64_t)(S - 1) * d : x + n * (int64_t)(S - 1) * d) + c;
    float* yr = y + n * (int64_t)S * d + c;
    float s = 0.f;
    for (int t = 0; t < S - 1; ++t) {
        const float v = xr[(int64_t)t * d];
        s += v;
        const float w = pos ? v + pos[(int64_t)(t + 1) * d + c] : v;
        yr[(int64_t)(t + 1) * d] = w;
        if (packed) packed[p1_offset(n * S + t + 1, c, KBp)] = (__bf16)w;       // bf16 mode: layer 0's packed A operand
    }
    float cv = cls ? cls[c] : s / (float)(S - 1);
    if (pos) cv += pos[c];
    yr[0] = cv;
    if (packed) packed[p1_offset(n * S, c, KBp)] = (__bf16)cv;
}

// The same with FOUR columns per thread (d a multiple of 4, 16-B aligned operands): 16-B loads / stores, 8-B pack stores, the
// token loop unrolled so that several rows are in flight per lane.  Same per-column arithmetic in the same order (the mean CLS
// token is a sequential sum over the tokens), so the results are bitwise those of the scalar kernel, which took 569 us for the
// headline batch in bf16 mode (2.0 GB moved: 3.6 TB/s) with its 4-B loads and 2-B pack stores.
// GATHER (round 5): ``clip_idx`` != NULL makes x a feature BANK [clips, P, d] and token t of sequence n the patch t % P of clip
// clip_idx[n * (S - 1) / P + t / P] - the batch formation (lstc_gather_rows: 805 MB written and read again at the headline shape) is
// fused into this pass.  The index of a token is wave-uniform (one sequence per workgroup): a scalar load.
template <bool GATHER>
__global__ void __launch_bounds__(NT) cls_concat_fwd_vec4_kernel(const float4* __restrict__ x, const float4* __restrict__ x_hi,
                                                                  int64_t n_lo, const float4* __restrict__ cls,
                                                                  const float4* __restrict__ pos, float4* __restrict__ y, int S,
                                                                  int d4, __bf16* __restrict__ packed, int KBp,
                                                                  const int64_t* __restrict__ clip_idx, int P) {
    const int64_t n = blockIdx.x;
    const int c4 = blockIdx.y * NT + threadIdx.x;
    if (c4 >= d4) return;
    const float4* xr = (x_hi && n >= n_lo ? x_hi + (n - n_lo) * (int64_t)(S - 1) * d4 : x + n * (int64_t)(S - 1) * d4) + c4;
    const int64_t* ci = GATHER ? clip_idx + n * (int64_t)((S - 1) / P) : nullptr;
    float4* yr = y ? y + n * (int64_t)S * d4 + c4 : nullptr;       // NULL: pack only (bf16 activation stream)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    auto emit = [&](int64_t row, const float4& w) {
        bf16x4v h;
        h[0] = (__bf16)w.x; h[1] = (__bf16)w.y; h[2] = (__bf16)w.z; h[3] = (__bf16)w.w;
        *reinterpret_cast<bf16x4v*>(packed + p1_offset(row, 4 * c4, KBp)) = h;
    };
#ifndef CLS_CONCAT_UNROLL
#define CLS_CONCAT_UNROLL 8      /* loads in flight per thread: 4 -> 8 took the kernel 319 -> 286 us (f32 only), 480 -> 395 us (with the pack); 16: 303 / 387 */
#endif
#pragma unroll CLS_CONCAT_UNROLL
    for (int t = 0; t < S - 1; ++t) {
        float4 v;
        if constexpr (GATHER) v = x[(ci[t / P] * P + t % P) * (int64_t)d4 + c4];
        else v = xr[(int64_t)t * d4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        float4 w = v;
        if (pos) {
            const float4 pv = pos[(int64_t)(t + 1) * d4 + c4];
            w.x += pv.x; w.y += pv.y; w.z += pv.z; w.w += pv.w;
        }
        if (yr) yr[(int64_t)(t + 1) * d4] = w;
        if (packed) emit(n * S + t + 1, w);
    }
    float4 cv;
    if (cls) cv = cls[c4];
    else { const float r = (float)(S - 1); cv = make_float4(s.x / r, s.y / r, s.z / r, s.w / r); }
    if (pos) { const float4 pv = pos[c4]; cv.x += pv.x; cv.y += pv.y; cv.z += pv.z; cv.w += pv.w; }
    if (yr) yr[0] = cv;
    if (packed) emit(n * S, cv);
}

// dx[n,t,:] = dy[n,t+1,:] + (mean_cls ? dy[n,0,:]/(S-1) : 0)
__global__ void __launch_bounds__(NT) cls_concat_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int S,
                                                             int d, int mean_cls) {
    const int64_t n = blockIdx.x;
    const int c = blockIdx.y * NT + threadIdx.x;
    if (c >= d) return;
    const float* dr = dy + n * (int64_t)S * d + c;
    float* xr = dx + n * (int64_t)(S - 1) * d + c;
    const float g0 = mean_cls ? dr[0] / (float)(S - 1) : 0.f;
    for (int t = 0; t < S - 1; ++t) xr[(int64_t)t * d] = dr[(int64_t)(t + 1) * d] + g0;
}

// ---------------------------------------------------------------------------------- colsum
// pass 1: grid (ceil(cols/NT), n_partial): workgroup y sums rows y, y+n_partial, ... of NT columns.
// (blockIdx.z = plane of a batched call, lstc_colsum_batched: planes are `xstride` / n_partial * cols floats apart)
__global__ void __launch_bounds__(NT) colsum_pass1(const float* __restrict__ x, int64_t rows, int cols, int ld,
                                                    float* __restrict__ partial, int64_t xstride = 0) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= cols) return;
    x += (size_t)blockIdx.z * xstride;
    partial += (size_t)blockIdx.z * gridDim.y * cols;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int64_t step = gridDim.y;
    int64_t r = blockIdx.y;
    for (; r + 3 * step < rows; r += 4 * step) {
        s0 += x[r * ld + c];
        s1 += x[(r + step) * ld + c];
        s2 += x[(r + 2 * step) * ld + c];
        s3 += x[(r + 3 * step) * ld + c];
    }
    for (; r < rows; r += step) s0 += x[r * ld + c];
    partial[(size_t)blockIdx.y * cols + c] = (s0 + s1) + (s2 + s3);
}
// pass 2: 64 threads per column group of 64 columns x 4 partial-row lanes; each thread sums every 4th partial row with
// 4 independent accumulators (the loop is latency-bound, not bandwidth-bound), then the 4 lanes combine through LDS.
__global__ void __launch_bounds__(NT) colsum_pass2(const float* __restrict__ partial, int n_partial, int cols,
                                                    float* __restrict__ out, int accumulate) {
    __shared__ float red[NT];
    partial += (size_t)blockIdx.y * n_partial * cols;          // blockIdx.y = plane of a batched call (0 otherwise)
    out += (size_t)blockIdx.y * cols;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int lane_p = threadIdx.x >> 6;                 // 0..3
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int p = lane_p;
        for (; p + 12 < n_partial; p += 16) {
            s0 += partial[(size_t)p * cols + c];
            s1 += partial[(size_t)(p + 4) * cols + c];
            s2 += partial[(size_t)(p + 8) * cols + c];
            s3 += partial[(size_t)(p + 12) * cols + c];
        }
        for (; p < n_partial; p += 4) s0 += partial[(size_t)p * cols + c];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (lane_p == 0 && c < cols) {
        const float s = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
        out[c] = accumulate ? out[c] + s : s;
    }
}

// A few rows (the partial products of a split-K weight gradient: 2-8 rows of out*in columns): ONE pass, float4 columns, the
// summation order of pass 1 + pass 2 on the same input (row p goes to accumulator p & 3 in increasing p, then (a0 + a1) +
// (a2 + a3)), so the result is bit-identical to the two-pass form - which copied the rows once before adding them.
__global__ void __launch_bounds__(NT) colsum_few(const float* __restrict__ x, int rows, int cols, int ld, float* __restrict__ out,
                                                  int accumulate) {
    const int c4 = (blockIdx.x * NT + threadIdx.x) * 4;
    if (c4 >= cols) return;
    float4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r0 = 0; r0 < rows; r0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + j < rows) {
                const float4 v = *reinterpret_cast<const float4*>(x + (size_t)(r0 + j) * ld + c4);
                a[j].x += v.x; a[j].y += v.y; a[j].z += v.z; a[j].w += v.w;
            }
    }
    float4 s;
    s.x = (a[0].x + a[1].x) + (a[2].x + a[3].x);
    s.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
    s.z = (a[0].z + a[1].z) + (a[2].z + a[3].z);
    s.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
    float4* o = reinterpret_cast<float4*>(out + c4);
    if (accumulate) { const float4 q = *o; s.x = q.x + s.x; s.y = q.y + s.y; s.z = q.z + s.z; s.w = q.w + s.w; }
    *o = s;
}

// C = epi(parts[0] + parts[1] + ... + parts[s - 1]) with the epilogue of lstc_gemm (csrc/gemm_f32.hip: bias, ReLU, dropout of the flat
// index row * N + col, residual, ReLU mask, accumulate - in that order); the K chunks of a small product arrive as separate partial
// results and are added in chunk order (no atomics).  One thread per four columns.
// `groups` > 1 (per-head products): group g's partials start at parts + g * group_stride_parts, its [M, N] result at C + g * group_stride_c.
__global__ void __launch_bounds__(NT) splitk_finish_kernel(const float* __restrict__ parts, int s, int64_t part_stride, int M, int N,
                                                            const float* __restrict__ bias, const float* __restrict__ res, int64_t ldr,
                                                            const float* __restrict__ relu_src, int64_t ld_relu, float* __restrict__ C,
                                                            int64_t ldc, int flags, DropKey dk, int groups, int64_t group_stride_parts,
                                                            int64_t group_stride_c) {
    dk = drop_key_now(dk);
    const int n4 = N >> 2;
    const int64_t per_group = (int64_t)M * n4, total = per_group * groups;
    for (int64_t q0 = (int64_t)blockIdx.x * NT + threadIdx.x; q0 < total; q0 += (int64_t)gridDim.x * NT) {
        const int grp = (int)(q0 / per_group);
        const int64_t q = q0 - (int64_t)grp * per_group;
        const int row = (int)(q / n4), col = 4 * (int)(q - (int64_t)row * n4);
        const float* pp = parts + (int64_t)grp * group_stride_parts + (int64_t)row * N + col;
        float4 v = *reinterpret_cast<const float4*>(pp);
        for (int i = 1; i < s; ++i) {
            const float4 x = *reinterpret_cast<const float4*>(pp + i * part_stride);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
        if (flags & LSTC_EPI_BIAS) {
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (flags & LSTC_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (flags & LSTC_EPI_DROPOUT) {
            const uint32_t idx = (uint32_t)row * (uint32_t)N + (uint32_t)col;
            v.x = drop_keep(idx, dk) ? v.x * dk.scale : 0.f;
            v.y = drop_keep(idx + 1, dk) ? v.y * dk.scale : 0.f;
            v.z = drop_keep(idx + 2, dk) ? v.z * dk.scale : 0.f;
            v.w = drop_keep(idx + 3, dk) ? v.w * dk.scale : 0.f;
        }
        if (flags & LSTC_EPI_RESIDUAL) {
            const float4 x = *reinterpret_cast<const float4*>(res + (int64_t)row * ldr + col);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
        if (flags & LSTC_EPI_RELU_MASK) {
            const float4 x = *reinterpret_cast<const float4*>(relu_src + (int64_t)row * ld_relu + col);
            v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f; v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
        }
        float4* cp = reinterpret_cast<float4*>(C + (int64_t)grp * group_stride_c + (int64_t)row * ldc + col);
        if (flags & LSTC_EPI_ACCUM) { const float4 x = *cp; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
        *cp = v;
    }
}

// out[i, :] = widen(pack row row0 + i * step): n rows of an lstc_pack1 operand back as f32 (one thread per 16-B chunk)
__global__ void __launch_bounds__(NT) unpack1_rows_kernel(const __bf16* __restrict__ pk, int KBp, int64_t row0, int64_t step, int64_t n,
                                                           int K, float* __restrict__ out, int64_t ldo) {
    const int cpr = K >> 3;
    const int64_t total = n * cpr;
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < total; q += (int64_t)gridDim.x * NT) {
        const int64_t i = q / cpr;
        const int c = (int)(q - i * cpr);
        const bf16x8v h = *reinterpret_cast<const bf16x8v*>(pk + p1_offset(row0 + i * step, 8 * c, KBp));
        float4* o = reinterpret_cast<float4*>(out + i * ldo + 8 * c);
        o[0] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        o[1] = make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]);
    }
}

// --------------------------------------------------------------------------------- dropout
__global__ void __launch_bounds__(NT) dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int64_t n, DropKey k) {
    k = drop_key_now(k);
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride)
        y[i] = drop_keep((uint32_t)i, k) ? x[i] * k.scale : 0.f;
}
// The same on an lstc_pack1 operand [rows, d] (bf16 activation stream: the backward of a block WITHOUT LayerNorm, whose incoming
// gradient pack is the gradient of dropout(f) + x as it stands).  A thread owns one 16-B chunk in STORAGE order (fully coalesced);
// the element's flat index row * d + col - the mask's counter - follows from the chunk's place in its 128-row x 32-k tile.
__global__ void __launch_bounds__(NT) dropout_apply_pack_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int64_t chunks,
                                                                 int KBp, int d, DropKey k) {
    k = drop_key_now(k);
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t q = (int64_t)blockIdx.x * NT + threadIdx.x; q < chunks; q += stride) {
        const int64_t tile = q >> 9;
        const int rr = (int)(q & 511) >> 2, pc = (int)(q & 3), ch = pc ^ ((rr >> 2) & 3);
        const int64_t row = (tile / KBp) * 128 + rr;
        const int col0 = (int)(tile % KBp) * 32 + ch * 8;
        const uint32_t fi = (uint32_t)(row * d) + (uint32_t)col0;
        const bf16x8v v = reinterpret_cast<const bf16x8v*>(x)[q];
        bf16x8v o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = drop_keep(fi + j, k) ? (__bf16)((float)v[j] * k.scale) : (__bf16)0.f;
        reinterpret_cast<bf16x8v*>(y)[q] = o;
    }
}
__global__ void __launch_bounds__(NT) dropout_mask_kernel(uint8_t* __restrict__ m, int64_t n, DropKey k) {
    k = drop_key_now(k);
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) m[i] = drop_keep((uint32_t)i, k) ? 1 : 0;
}

// ------------------------------------------------------------------------------ head output
// x [rows,32] -> c logits -> sigmoid (c=1) or softmax (c=2); thread per row.
__global__ void __launch_bounds__(NT) head_out_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ b, float* __restrict__ out,
                                                           int64_t rows, int c) {
    __shared__ float w[2 * 32 + 2];
    if (threadIdx.x < c * 32) w[threadIdx.x] = W[threadIdx.x];
    if (threadIdx.x < c) w[64 + threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (r >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + r * 32);
    float z0 = 0.f, z1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 v = xr[i];
        z0 += v.x * w[4 * i] + v.y * w[4 * i + 1] + v.z * w[4 * i + 2] + v.w * w[4 * i + 3];
        if (c == 2) z1 += v.x * w[32 + 4 * i] + v.y * w[32 + 4 * i + 1] + v.z * w[32 + 4 * i + 2] + v.w * w[32 + 4 * i + 3];
    }
    z0 += w[64];
    if (c == 1) {
        out[r] = 1.f / (1.f + expf(-z0));
    } else {
        z1 += w[65];
        const float m = fmaxf(z0, z1);
        const float e0 = expf(z0 - m), e1 = expf(z1 - m);
        const float inv = 1.f / (e0 + e1);
        out[2 * r] = e0 * inv;
        out[2 * r + 1] = e1 * inv;
    }
}

// ONE workgroup walks all rows (a few thousand at most: one row per sequence): every thread keeps its own partial sums of
// dW / db in registers, then wave shuffles and a wave-ordered LDS sum - a fixed summation order, no atomics.
__global__ void __launch_bounds__(NT) head_out_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ out, const float* __restrict__ dout,
                                                           float* __restrict__ dx, float* __restrict__ dW,
                                                           float* __restrict__ db, int64_t rows, int c) {
    __shared__ float w[64];
    __shared__ float accw[NT / 64][2 * 32 + 2];
    if (threadIdx.x < c * 32) w[threadIdx.x] = W[threadIdx.x];
    __syncthreads();
    float a0[32], a1[32], b0 = 0.f, b1 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) a0[i] = a1[i] = 0.f;
    for (int64_t r = threadIdx.x; r < rows; r += NT) {
        float dz0, dz1 = 0.f;
        if (c == 1) {
            const float o = out[r];
            dz0 = dout[r] * o * (1.f - o);
        } else {
            const float o0 = out[2 * r], o1 = out[2 * r + 1], g0 = dout[2 * r], g1 = dout[2 * r + 1];
            const float dot = g0 * o0 + g1 * o1;
            dz0 = o0 * (g0 - dot);
            dz1 = o1 * (g1 - dot);
        }
        const float4* xr = reinterpret_cast<const float4*>(x + r * 32);
        float4* dr = reinterpret_cast<float4*>(dx + r * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 v = xr[i];
            float4 d4;
            d4.x = dz0 * w[4 * i] + (c == 2 ? dz1 * w[32 + 4 * i] : 0.f);
            d4.y = dz0 * w[4 * i + 1] + (c == 2 ? dz1 * w[32 + 4 * i + 1] : 0.f);
            d4.z = dz0 * w[4 * i + 2] + (c == 2 ? dz1 * w[32 + 4 * i + 2] : 0.f);
            d4.w = dz0 * w[4 * i + 3] + (c == 2 ? dz1 * w[32 + 4 * i + 3] : 0.f);
            dr[i] = d4;
            a0[4 * i] += dz0 * v.x; a0[4 * i + 1] += dz0 * v.y; a0[4 * i + 2] += dz0 * v.z; a0[4 * i + 3] += dz0 * v.w;
            a1[4 * i] += dz1 * v.x; a1[4 * i + 1] += dz1 * v.y; a1[4 * i + 2] += dz1 * v.z; a1[4 * i + 3] += dz1 * v.w;
        }
        b0 += dz0; b1 += dz1;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float s0 = wave_sum(a0[i]), s1 = wave_sum(a1[i]);
        if (lane == 0) { accw[wave][i] = s0; accw[wave][32 + i] = s1; }
    }
    {
        const float s0 = wave_sum(b0), s1 = wave_sum(b1);
        if (lane == 0) { accw[wave][64] = s0; accw[wave][65] = s1; }
    }
    __syncthreads();
    if (threadIdx.x < 66) {
        float v = accw[0][threadIdx.x];
#pragma unroll
        for (int q = 1; q < NT / 64; ++q) v += accw[q][threadIdx.x];
        if (threadIdx.x < c * 32) dW[threadIdx.x] = v;             // written, not added to: one workgroup holds the whole sum
        else if (threadIdx.x >= 64 && threadIdx.x - 64 < c) db[threadIdx.x - 64] = v;
    }
}

// --------------------------------------------------------------------------------- Adagrad
__global__ void __launch_bounds__(NT) adagrad_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                      float* __restrict__ s, int64_t n, float lr, float wd, float eps,
                                                      float gscale) {
    const int64_t stride = (int64_t)gridDim.x * NT;
    const int64_t n4 = n >> 2;
    const bool vec = aligned16(w) && aligned16(g) && aligned16(s);
    int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (vec) {
        for (; i < n4; i += stride) {
            float4 wv = reinterpret_cast<float4*>(w)[i], sv = reinterpret_cast<float4*>(s)[i];
            const float4 gv = reinterpret_cast<const float4*>(g)[i];
            float gg;
            gg = gv.x * gscale + wd * wv.x; sv.x += gg * gg; wv.x -= lr * gg / (sqrtf(sv.x) + eps);
            gg = gv.y * gscale + wd * wv.y; sv.y += gg * gg; wv.y -= lr * gg / (sqrtf(sv.y) + eps);
            gg = gv.z * gscale + wd * wv.z; sv.z += gg * gg; wv.z -= lr * gg / (sqrtf(sv.z) + eps);
            gg = gv.w * gscale + wd * wv.w; sv.w += gg * gg; wv.w -= lr * gg / (sqrtf(sv.w) + eps);
            reinterpret_cast<float4*>(w)[i] = wv;
            reinterpret_cast<float4*>(s)[i] = sv;
        }
        i = n4 * 4 + (int64_t)blockIdx.x * NT + threadIdx.x;
    }
    for (; i < n; i += stride) {
        const float gg = g[i] * gscale + wd * w[i];
        const float sv = s[i] + gg * gg;
        s[i] = sv;
        w[i] -= lr * gg / (sqrtf(sv) + eps);
    }
}

// All parameters of an optimizer step in ONE launch: up to ADA_MAX tensors ride in the kernel arguments (no device-side table,
// no host-to-device copy); a workgroup finds its tensor by scanning the prefix of workgroup counts and then runs the same
// element update as adagrad_kernel on its slice.  Element arithmetic and order are adagrad_kernel's: results are bit-identical.
constexpr int ADA_MAX = 48;
constexpr int ADA_PER_WG = NT * 4 * 8;        // elements per workgroup: 8 float4 per thread
struct AdaBatch {
    float* w[ADA_MAX];
    const float* g[ADA_MAX];
    float* s[ADA_MAX];
    long long n[ADA_MAX];
    float lr[ADA_MAX], wd[ADA_MAX], eps[ADA_MAX], gscale[ADA_MAX];
    int first_wg[ADA_MAX + 1];
    int count;
};
__global__ void __launch_bounds__(NT) adagrad_multi_kernel(const AdaBatch b) {
    int t = 0;
    while (t + 1 < b.count && (int)blockIdx.x >= b.first_wg[t + 1]) ++t;          // wave-uniform scan (scalar loads of the arguments)
    float* __restrict__ w = b.w[t];
    const float* __restrict__ g = b.g[t];
    float* __restrict__ s = b.s[t];
    const int64_t n = b.n[t];
    const float lr = b.lr[t], wd = b.wd[t], eps = b.eps[t], gscale = b.gscale[t];
    const int64_t e0 = (int64_t)((int)blockIdx.x - b.first_wg[t]) * ADA_PER_WG;
    const int64_t e1 = min(n, e0 + ADA_PER_WG);
    const bool vec = aligned16(w) && aligned16(g) && aligned16(s);
    if (vec) {
        const int64_t v1 = e1 == n ? (n >> 2) : (e1 >> 2);
        for (int64_t i = (e0 >> 2) + threadIdx.x; i < v1; i += NT) {
            float4 wv = reinterpret_cast<float4*>(w)[i], sv = reinterpret_cast<float4*>(s)[i];
            const float4 gv = reinterpret_cast<const float4*>(g)[i];
            float gg;
            gg = gv.x * gscale + wd * wv.x; sv.x += gg * gg; wv.x -= lr * gg / (sqrtf(sv.x) + eps);
            gg = gv.y * gscale + wd * wv.y; sv.y += gg * gg; wv.y -= lr * gg / (sqrtf(sv.y) + eps);
            gg = gv.z * gscale + wd * wv.z; sv.z += gg * gg; wv.z -= lr * gg / (sqrtf(sv.z) + eps);
            gg = gv.w * gscale + wd * wv.w; sv.w += gg * gg; wv.w -= lr * gg / (sqrtf(sv.w) + eps);
            reinterpret_cast<float4*>(w)[i] = wv;
            reinterpret_cast<float4*>(s)[i] = sv;
        }
        if (e1 != n) return;
        for (int64_t i = (n >> 2) * 4 + threadIdx.x; i < n; i += NT) {          // the tensor's last 0-3 elements
            const float gg = g[i] * gscale + wd * w[i];
            const float sv = s[i] + gg * gg;
            s[i] = sv;
            w[i] -= lr * gg / (sqrtf(sv) + eps);
        }
        return;
    }
    for (int64_t i = e0 + threadIdx.x; i < e1; i += NT) {
        const float gg = g[i] * gscale + wd * w[i];
        const float sv = s[i] + gg * gg;
        s[i] = sv;
        w[i] -= lr * gg / (sqrtf(sv) + eps);
    }
}

__global__ void __launch_bounds__(NT) sqnorm_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[NT / 64];
    const int64_t stride = (int64_t)gridDim.x * NT;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) s += x[i] * x[i];
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// clip_grad_norm_ over a parameter LIST in two launches and without a host read-back (the reference: one norm kernel per tensor, a
// stack, a norm of norms and a python comparison, Train/temporal_transformer_shanghaitech.py:139-141).  Items ride in the kernel
// arguments like adagrad_multi_kernel's; workgroup w owns one ADA_PER_WG-element slice of one tensor and writes ONE partial
// (fixed slice -> workgroup map, fixed tree inside the workgroup), a single workgroup then adds the partials in index order:
// the norm is run-to-run bit-identical (no float atomics).
struct VecBatch {
    float* x[ADA_MAX];
    long long n[ADA_MAX];
    int first_wg[ADA_MAX + 1];
    int count;
};
__global__ void __launch_bounds__(NT) sqnorm_multi_kernel(const VecBatch b, float* __restrict__ partials) {
    __shared__ float red[NT / 64];
    int t = 0;
    while (t + 1 < b.count && (int)blockIdx.x >= b.first_wg[t + 1]) ++t;
    const float* __restrict__ x = b.x[t];
    const int64_t n = b.n[t];
    const int64_t e0 = (int64_t)((int)blockIdx.x - b.first_wg[t]) * ADA_PER_WG;
    const int64_t e1 = min(n, e0 + ADA_PER_WG);
    float s = 0.f;
    if (aligned16(x)) {          // e0 is a multiple of 4: float4 over the whole quads, then the tensor's last 0-3 elements
        const int64_t v1 = e1 >> 2;
        for (int64_t i = (e0 >> 2) + threadIdx.x; i < v1; i += NT) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            s += v.x * v.x; s += v.y * v.y; s += v.z * v.z; s += v.w * v.w;
        }
        for (int64_t i = (v1 << 2) + threadIdx.x; i < e1; i += NT) s += x[i] * x[i];
    } else {
        for (int64_t i = e0 + threadIdx.x; i < e1; i += NT) s += x[i] * x[i];
    }
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ void __launch_bounds__(NT) sum_partials_kernel(const float* __restrict__ partials, int64_t n, float* __restrict__ out) {
    __shared__ float red[NT / 64];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += NT) s += partials[i];
    s = block_sum<NT>(s, red);
    if (threadIdx.x == 0) { out[0] = s; out[1] = sqrtf(s); }
}
// x *= min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) for every tensor of the list, the coefficient formed on the device
// (torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1); coefficient 1 leaves the data untouched.
// A NaN total norm gives a NaN coefficient, and torch.clamp keeps NaN: every gradient is multiplied by it there - so here too
// (round 5; rounds 1-4 returned early and kept a diverged step's NaN local).
__global__ void __launch_bounds__(NT) clip_scale_multi_kernel(const VecBatch b, const float* __restrict__ sqnorm, float max_norm) {
    const float coef = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
    if (coef >= 1.0f) return;                 // NaN falls through: the gradients become NaN, as in torch
    int t = 0;
    while (t + 1 < b.count && (int)blockIdx.x >= b.first_wg[t + 1]) ++t;
    float* __restrict__ x = b.x[t];
    const int64_t n = b.n[t];
    const int64_t e0 = (int64_t)((int)blockIdx.x - b.first_wg[t]) * ADA_PER_WG;
    const int64_t e1 = min(n, e0 + ADA_PER_WG);
    if (aligned16(x)) {
        const int64_t v1 = e1 >> 2;
        for (int64_t i = (e0 >> 2) + threadIdx.x; i < v1; i += NT) {
            float4 v = reinterpret_cast<float4*>(x)[i];
            v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
            reinterpret_cast<float4*>(x)[i] = v;
        }
        for (int64_t i = (v1 << 2) + threadIdx.x; i < e1; i += NT) x[i] *= coef;
    } else {
        for (int64_t i = e0 + threadIdx.x; i < e1; i += NT) x[i] *= coef;
    }
}

__global__ void __launch_bounds__(NT) scale_kernel(float* __restrict__ x, int64_t n, float alpha) {
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) x[i] *= alpha;
}

// f32 <-> bf16 (RNE) streams for a half-width gradient all-reduce (lstc_vad_amd/dist.py, reduce_dtype="bf16")
__global__ void __launch_bounds__(NT) cast_f32_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) y[i] = (__bf16)x[i];
}
__global__ void __launch_bounds__(NT) cast_bf16_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) y[i] = (float)x[i];
}

// dst row r <- src row idx[r].  grid.x = row, grid.y = chunk of the row; float4 streaming copy (HBM-bound, 2 x bytes).
__global__ void __launch_bounds__(NT) gather_rows_kernel(const float4* __restrict__ src, int64_t src_rows,
                                                         const int64_t* __restrict__ idx, float4* __restrict__ dst,
                                                         int64_t row_vec) {
    const int64_t r = blockIdx.x;
    int64_t s = idx[r];
    s = s < 0 ? 0 : (s >= src_rows ? src_rows - 1 : s);          // never fault on a bad index
    const float4* __restrict__ in = src + s * row_vec;
    float4* __restrict__ out = dst + r * row_vec;
    for (int64_t i = (int64_t)blockIdx.y * NT + threadIdx.x; i < row_vec; i += (int64_t)gridDim.y * NT)
        out[i] = in[i];
}

inline int grid_for(int64_t work_items, int per_block, int cap = 2048) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

extern "C" {

static int ln_fwd_launch(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                         int64_t rows, int32_t d, float eps, void* packed, int KBp, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return LSTC_E_NULL;
    if (rows <= 0 || d <= 0) return LSTC_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(rows, NT / 64, 4096);
    const bool vec = d % 4 == 0 && d <= 2048 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta);
    if (vec) {
        if (d <= 256) hipLaunchKernelGGL(ln_fwd_vec<1>, grid, NT, 0, st, x, gamma, beta, y, mean, rstd, rows, d, eps, (__bf16*)packed, KBp);
        else if (d <= 512) hipLaunchKernelGGL(ln_fwd_vec<2>, grid, NT, 0, st, x, gamma, beta, y, mean, rstd, rows, d, eps, (__bf16*)packed, KBp);
        else if (d <= 1024) hipLaunchKernelGGL(ln_fwd_vec<4>, grid, NT, 0, st, x, gamma, beta, y, mean, rstd, rows, d, eps, (__bf16*)packed, KBp);
        else hipLaunchKernelGGL(ln_fwd_vec<8>, grid, NT, 0, st, x, gamma, beta, y, mean, rstd, rows, d, eps, (__bf16*)packed, KBp);
    } else {
        if (packed) return LSTC_E_UNSUPPORTED;
        hipLaunchKernelGGL(ln_fwd_generic, grid, NT, 0, st, x, gamma, beta, y, mean, rstd, rows, d, eps);
    }
    return lstc_launch_status();
}

// The row-wise producers write whole 128-row x 32-k tiles only when the matrix fills the even-by-even tile grid of
// lstc_pack1 exactly (the GEMM streams tile pairs and would add the garbage of an unwritten tile into real outputs).
static bool pack_fused_ok(int64_t rows, int32_t d) { return rows % 256 == 0 && d % 64 == 0 && d <= 2048; }

int lstc_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int64_t rows, int32_t d, float eps, void* stream) {
    return ln_fwd_launch(x, gamma, beta, y, mean, rstd, rows, d, eps, nullptr, 0, stream);
}

int lstc_layernorm_fwd_pack(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            int64_t rows, int32_t d, float eps, void* packed, void* stream) {
    if (!packed) return LSTC_E_NULL;
    if (rows > 0 && d > 0 && !pack_fused_ok(rows, d)) return LSTC_E_UNSUPPORTED;
    if (!aligned16(packed)) return LSTC_E_ALIGN;
    return ln_fwd_launch(x, gamma, beta, y, mean, rstd, rows, d, eps, packed, d / 32, stream);
}

int lstc_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx, float* partial, int32_t n_partial, int64_t rows, int32_t d, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !partial) return LSTC_E_NULL;
    if (rows <= 0 || d <= 0 || n_partial <= 0) return LSTC_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = d % 4 == 0 && d <= 2048 && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) &&
                     aligned16(partial);
    if (vec) {
        // model widths: two waves per row (ln_bwd_pack2, 3 waves/SIMD); other widths: one wave per row
        const size_t lds = (size_t)(NT / 64) * d * sizeof(float), lds2 = (size_t)(2 * d + 1024) * sizeof(float);
        const DropKey k0 = make_drop_key(0.f, 0);
        __bf16* np = nullptr;
        if (d == 512) hipLaunchKernelGGL((ln_bwd_pack2<1, 0>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, np, 0, k0);
        else if (d == 1024) hipLaunchKernelGGL((ln_bwd_pack2<2, 0>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, np, 0, k0);
        else if (d == 2048) hipLaunchKernelGGL((ln_bwd_pack2<4, 0>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, np, 0, k0);
        else if (d <= 256) hipLaunchKernelGGL((ln_bwd_vec<1, false>), n_partial, NT, lds, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, np, 0, k0);
        else if (d <= 512) hipLaunchKernelGGL((ln_bwd_vec<2, false>), n_partial, NT, lds, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, np, 0, k0);
        else if (d <= 1024) hipLaunchKernelGGL((ln_bwd_vec<4, false>), n_partial, NT, lds, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, np, 0, k0);
        else hipLaunchKernelGGL((ln_bwd_vec<8, false>), n_partial, NT, lds, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, np, 0, k0);
    } else {
        if ((size_t)2 * d * sizeof(float) > 64 * 1024) return LSTC_E_RANGE;
        hipLaunchKernelGGL(ln_bwd_generic, n_partial, NT, (size_t)2 * d * sizeof(float), st, dy, x, gamma, mean, rstd, dx,
                           partial, rows, d);
    }
    return lstc_launch_status();
}

int lstc_layernorm_bwd_drop_pack(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 float* dx, float* partial, int32_t n_partial, int64_t rows, int32_t d, float dropout_p,
                                 uint64_t dropout_seed, void* packed, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !partial || !packed) return LSTC_E_NULL;
    if (rows <= 0 || d <= 0 || n_partial <= 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return LSTC_E_SHAPE;
    if (!pack_fused_ok(rows, d)) return LSTC_E_UNSUPPORTED;
    if ((uint64_t)rows * (uint64_t)d > 0xffffffffull) return LSTC_E_RANGE;      // 32-bit dropout counter
    if (!(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(gamma) && aligned16(partial) && aligned16(packed)))
        return LSTC_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const DropKey key = make_drop_key(dropout_p, dropout_seed);
    __bf16* pk = (__bf16*)packed;
    const int KBp = d / 32;
    // d = 512 / 1024 / 2048 (the model widths): two waves per row at 3 waves/SIMD; its column-to-lane map and summation
    // order equal ln_bwd_vec's exactly at these widths.  Other widths: one wave per row.
    const size_t lds2 = (size_t)(2 * d + 1024) * sizeof(float), lds1 = (size_t)(NT / 64) * d * sizeof(float);
    if (d == 512) hipLaunchKernelGGL((ln_bwd_pack2<1, 1>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, pk, KBp, key);
    else if (d == 1024) hipLaunchKernelGGL((ln_bwd_pack2<2, 1>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, pk, KBp, key);
    else if (d == 2048) hipLaunchKernelGGL((ln_bwd_pack2<4, 1>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, pk, KBp, key);
    else if (d <= 256) hipLaunchKernelGGL((ln_bwd_vec<1, true>), n_partial, NT, lds1, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, pk, KBp, key);
    else if (d <= 512) hipLaunchKernelGGL((ln_bwd_vec<2, true>), n_partial, NT, lds1, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, pk, KBp, key);
    else if (d <= 1024) hipLaunchKernelGGL((ln_bwd_vec<4, true>), n_partial, NT, lds1, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, pk, KBp, key);
    else hipLaunchKernelGGL((ln_bwd_vec<8, true>), n_partial, NT, lds1, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, pk, KBp, key);
    return lstc_launch_status();
}

int lstc_layernorm_bwd_drop(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                            float* dx, float* df, float* partial, int32_t n_partial, int64_t rows, int32_t d, float dropout_p,
                            uint64_t dropout_seed, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !df || !partial) return LSTC_E_NULL;
    if (rows <= 0 || d <= 0 || n_partial <= 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return LSTC_E_SHAPE;
    if (d != 512 && d != 1024 && d != 2048) return LSTC_E_UNSUPPORTED;          // the two-waves-per-row kernel's widths
    if ((uint64_t)rows * (uint64_t)d > 0xffffffffull) return LSTC_E_RANGE;      // 32-bit dropout counter
    if (!(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(df) && aligned16(gamma) && aligned16(partial))) return LSTC_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const DropKey key = make_drop_key(dropout_p, dropout_seed);
    const size_t lds2 = (size_t)(2 * d + 1024) * sizeof(float);
    __bf16* out = reinterpret_cast<__bf16*>(df);
    if (d == 512) hipLaunchKernelGGL((ln_bwd_pack2<1, 2>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, out, 0, key);
    else if (d == 1024) hipLaunchKernelGGL((ln_bwd_pack2<2, 2>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, out, 0, key);
    else hipLaunchKernelGGL((ln_bwd_pack2<4, 2>), n_partial, NT, lds2, st, dy, x, gamma, mean, rstd, dx, partial, rows, d, out, 0, key);
    return lstc_launch_status();
}

// bf16 activation stream: the model widths only (d = 1024 / 2048), rows filling the pack's tile grid
static bool act_shape_ok(int64_t rows, int32_t d) { return rows % 256 == 0 && (d == 1024 || d == 2048); }

int lstc_layernorm_fwd_act(const float* x, const void* x_pack, const float* gamma, const float* beta, float* y, void* y_pack,
                           float* mean, float* rstd, int64_t rows, int32_t d, float eps, void* stream) {
    if ((!x) == (!x_pack) || !gamma || !beta || (!y && !y_pack) || !mean || !rstd) return LSTC_E_NULL;
    if (rows <= 0 || d <= 0) return LSTC_E_SHAPE;
    if (!act_shape_ok(rows, d)) return LSTC_E_UNSUPPORTED;
    if (!(aligned16(x) && aligned16(x_pack) && aligned16(y) && aligned16(y_pack) && aligned16(gamma) && aligned16(beta))) return LSTC_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(rows, NT / 64, 4096);
    const int KBp = d / 32;
    const __bf16* xp = (const __bf16*)x_pack;
    __bf16* yp = (__bf16*)y_pack;
#define LN_FWD_ACT(NC, XP, YF, YP) hipLaunchKernelGGL((ln_fwd_act<NC, XP, YF, YP>), grid, NT, 0, st, x, xp, gamma, beta, y, yp, mean, rstd, rows, d, eps, KBp)
#define LN_FWD_ACT_D(XP, YF, YP) do { if (d == 1024) LN_FWD_ACT(2, XP, YF, YP); else LN_FWD_ACT(4, XP, YF, YP); } while (0)
    if (x_pack) {
        if (y && y_pack) LN_FWD_ACT_D(true, true, true);
        else if (y) LN_FWD_ACT_D(true, true, false);
        else LN_FWD_ACT_D(true, false, true);
    } else {
        if (y && y_pack) LN_FWD_ACT_D(false, true, true);
        else if (y) LN_FWD_ACT_D(false, true, false);
        else LN_FWD_ACT_D(false, false, true);
    }
#undef LN_FWD_ACT_D
#undef LN_FWD_ACT
    return lstc_launch_status();
}

int lstc_layernorm_bwd_act(const float* dy, const void* dy_pack, const void* x_pack, const float* gamma, const float* mean,
                           const float* rstd, void* dx_pack, float* partial, int32_t n_partial, int64_t rows, int32_t d,
                           float dropout_p, uint64_t dropout_seed, void* df_pack, void* stream) {
    if ((!dy) == (!dy_pack) || !x_pack || !gamma || !mean || !rstd || !partial || !df_pack) return LSTC_E_NULL;
    if (rows <= 0 || d <= 0 || n_partial <= 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) return LSTC_E_SHAPE;
    if (!act_shape_ok(rows, d)) return LSTC_E_UNSUPPORTED;
    if ((uint64_t)rows * (uint64_t)d > 0xffffffffull) return LSTC_E_RANGE;      // 32-bit dropout counter
    if (!(aligned16(dy) && aligned16(dy_pack) && aligned16(x_pack) && aligned16(dx_pack) && aligned16(gamma) && aligned16(partial) &&
          aligned16(df_pack))) return LSTC_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const DropKey key = make_drop_key(dropout_p, dropout_seed);
    const int KBp = d / 32;
    const size_t lds2 = (size_t)(2 * d + 1024) * sizeof(float);
    const __bf16 *dyp = (const __bf16*)dy_pack, *xp = (const __bf16*)x_pack;
    __bf16 *dxp = (__bf16*)dx_pack, *dfp = (__bf16*)df_pack;
#define LN_BWD_ACT(NCH, DYP, DX) hipLaunchKernelGGL((ln_bwd_act<NCH, DYP, DX>), n_partial, NT, lds2, st, dy, dyp, xp, gamma, mean, rstd, dxp, partial, rows, d, dfp, KBp, key)
#define LN_BWD_ACT_D(DYP, DX) do { if (d == 1024) LN_BWD_ACT(1, DYP, DX); else LN_BWD_ACT(2, DYP, DX); } while (0)
    if (dy_pack) { if (dx_pack) LN_BWD_ACT_D(true, true); else LN_BWD_ACT_D(true, false); }
    else { if (dx_pack) LN_BWD_ACT_D(false, true); else LN_BWD_ACT_D(false, false); }
#undef LN_BWD_ACT_D
#undef LN_BWD_ACT
    return lstc_launch_status();
}

static void launch_cls_concat_fwd(const float* x, const float* x_hi, int64_t n_lo, const float* cls_token, const float* pos, float* y,
                                  int64_t N, int32_t S, int32_t d, __bf16* packed, int KBp, hipStream_t st) {
    if (d % 4 == 0 && aligned16(x) && aligned16(y) && (!x_hi || aligned16(x_hi)) && (!cls_token || aligned16(cls_token)) &&
        (!pos || aligned16(pos))) {
        hipLaunchKernelGGL(cls_concat_fwd_vec4_kernel<false>, dim3((unsigned)N, (d / 4 + NT - 1) / NT), NT, 0, st, (const float4*)x,
                           (const float4*)x_hi, n_lo, (const float4*)cls_token, (const float4*)pos, (float4*)y, S, d / 4, packed, KBp,
                           (const int64_t*)nullptr, 1);
        return;
    }
    hipLaunchKernelGGL(cls_concat_fwd_kernel, dim3((unsigned)N, (d + NT - 1) / NT), NT, 0, st, x, x_hi, n_lo, cls_token, pos, y, S, d,
                       packed, KBp);
}

int lstc_cls_concat_fwd(const float* x, const float* x_hi, int64_t n_lo, const float* cls_token, const float* pos,
                        float* y, int64_t N, int32_t S, int32_t d, void* stream) {
    if (!x || !y) return LSTC_E_NULL;
    if (N <= 0 || S < 2 || d <= 0 || (x_hi && (n_lo < 0 || n_lo > N))) return LSTC_E_SHAPE;
    launch_cls_concat_fwd(x, x_hi, n_lo, cls_token, pos, y, N, S, d, nullptr, 0, (hipStream_t)stream);
    return lstc_launch_status();
}

int lstc_cls_concat_fwd_pack(const float* x, const float* x_hi, int64_t n_lo, const float* cls_token, const float* pos,
                             float* y, int64_t N, int32_t S, int32_t d, void* packed, void* stream) {
    if (!x || !packed) return LSTC_E_NULL;
    if (N <= 0 || S < 2 || d <= 0 || (x_hi && (n_lo < 0 || n_lo > N))) return LSTC_E_SHAPE;
    if ((N * S) % 256 != 0 || d % 64 != 0) return LSTC_E_UNSUPPORTED;      // the rows fill the pack's even tile grid exactly
    if (!aligned16(packed)) return LSTC_E_ALIGN;
    // y == NULL (pack only: the bf16 activation stream) exists on the 16-B kernel
    if (!y && !(d % 4 == 0 && aligned16(x) && (!x_hi || aligned16(x_hi)) && (!cls_token || aligned16(cls_token)) && (!pos || aligned16(pos))))
        return LSTC_E_ALIGN;
    launch_cls_concat_fwd(x, x_hi, n_lo, cls_token, pos, y, N, S, d, (__bf16*)packed, d / 32, (hipStream_t)stream);
    return lstc_launch_status();
}

int lstc_cls_concat_gather_fwd(const float* bank, int64_t bank_clips, const int64_t* clip_idx, int32_t P, const float* cls_token,
                               const float* pos, float* y, int64_t N, int32_t S, int32_t d, void* packed, void* stream) {
    if (!bank || !clip_idx || (!y && !packed)) return LSTC_E_NULL;
    if (N <= 0 || S < 2 || d <= 0 || P <= 0 || bank_clips <= 0 || (S - 1) % P != 0) return LSTC_E_SHAPE;
    if (packed && ((N * S) % 256 != 0 || d % 64 != 0)) return LSTC_E_UNSUPPORTED;      // the rows fill the pack's even tile grid exactly
    if (d % 4 != 0 || !aligned16(bank) || !aligned16(y) || !aligned16(packed) || (cls_token && !aligned16(cls_token)) || (pos && !aligned16(pos)))
        return LSTC_E_ALIGN;
    hipLaunchKernelGGL(cls_concat_fwd_vec4_kernel<true>, dim3((unsigned)N, (d / 4 + NT - 1) / NT), NT, 0, (hipStream_t)stream,
                       (const float4*)bank, (const float4*)nullptr, (int64_t)0, (const float4*)cls_token, (const float4*)pos, (float4*)y, S,
                       d / 4, (__bf16*)packed, d / 32, clip_idx, P);
    return lstc_launch_status();
}

int lstc_cls_concat_bwd(const float* dy, float* dx, int64_t N, int32_t S, int32_t d, int32_t mean_cls, void* stream) {
    if (!dy || !dx) return LSTC_E_NULL;
    if (N <= 0 || S < 2 || d <= 0) return LSTC_E_SHAPE;
    hipLaunchKernelGGL(cls_concat_bwd_kernel, dim3((unsigned)N, (d + NT - 1) / NT), NT, 0, (hipStream_t)stream, dy, dx,
                       S, d, mean_cls);
    return lstc_launch_status();
}

int lstc_colsum(const float* x, int64_t rows, int32_t cols, int32_t ld, float* partial, int32_t n_partial, float* out,
                int32_t accumulate, void* stream) {
    if (!x || !partial || !out) return LSTC_E_NULL;
    if (rows <= 0 || cols <= 0 || ld < cols || n_partial <= 0) return LSTC_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int np = (int)(rows < n_partial ? rows : n_partial);
    if (rows <= 12 && rows <= n_partial && cols % 4 == 0 && ld % 4 == 0 && aligned16(x) && aligned16(out)) {
        hipLaunchKernelGGL(colsum_few, dim3((cols / 4 + NT - 1) / NT), NT, 0, st, x, (int)rows, cols, ld, out, accumulate);
        return lstc_launch_status();
    }
    hipLaunchKernelGGL(colsum_pass1, dim3((cols + NT - 1) / NT, np), NT, 0, st, x, rows, cols, ld, partial);
    hipLaunchKernelGGL(colsum_pass2, dim3((cols + 63) / 64), NT, 0, st, partial, np, cols, out, accumulate);
    return lstc_launch_status();
}

int lstc_colsum_batched(const float* x, int32_t batch, int64_t rows, int32_t cols, int32_t ld, int64_t batch_stride,
                        float* partial, int32_t n_partial, float* out, void* stream) {
    if (!x || !partial || !out) return LSTC_E_NULL;
    if (batch <= 0 || batch > 65535 || rows <= 0 || cols <= 0 || ld < cols || n_partial <= 0 || batch_stride < 0) return LSTC_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int np = (int)(rows < n_partial ? rows : n_partial);
    hipLaunchKernelGGL(colsum_pass1, dim3((cols + NT - 1) / NT, np, batch), NT, 0, st, x, rows, cols, ld, partial, batch_stride);
    hipLaunchKernelGGL(colsum_pass2, dim3((cols + 63) / 64, batch), NT, 0, st, partial, np, cols, out, 0);
    return lstc_launch_status();
}

int lstc_colsum_pack1(const void* packed, int64_t rows, int32_t K, float* partial, int32_t n_partial, float* out,
                      int32_t accumulate, void* stream) {
    if (!packed || !partial || !out) return LSTC_E_NULL;
    if (rows <= 0 || K <= 0 || n_partial <= 0) return LSTC_E_SHAPE;
    if (!aligned16(packed)) return LSTC_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rb = (rows + 127) / 128;
    const int RB = (int)(rb + (rb & 1)), KB = (K + 31) / 32, KBp = KB + (KB & 1);        // padded rows are zeros
    const int np = (int)(RB < n_partial ? RB : n_partial);
    hipLaunchKernelGGL(colsum_pack1_kernel, dim3(KB, np), NT, 0, st, (const __bf16*)packed, RB, KBp, K, partial);
    hipLaunchKernelGGL(colsum_pass2, dim3((K + 63) / 64), NT, 0, st, partial, np, K, out, accumulate);
    return lstc_launch_status();
}

int lstc_dropout_apply(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
    if (!x || !y) return LSTC_E_NULL;
    if (n <= 0) return LSTC_E_SHAPE;
    if ((uint64_t)n > 0xffffffffull) return LSTC_E_RANGE;
    hipLaunchKernelGGL(dropout_apply_kernel, grid_for(n, NT * 4), NT, 0, (hipStream_t)stream, x, y, n, make_drop_key(p, seed));
    return lstc_launch_status();
}

int lstc_dropout_apply_pack(const void* x_pack, void* y_pack, int64_t rows, int32_t d, float p, uint64_t seed, void* stream) {
    if (!x_pack || !y_pack) return LSTC_E_NULL;
    if (rows <= 0 || d <= 0 || !(p >= 0.f && p < 1.f)) return LSTC_E_SHAPE;
    if (rows % 256 != 0 || d % 64 != 0) return LSTC_E_UNSUPPORTED;            // the matrix fills the pack's even tile grid exactly
    if ((uint64_t)rows * (uint64_t)d > 0xffffffffull) return LSTC_E_RANGE;    // 32-bit dropout counter
    if (!aligned16(x_pack) || !aligned16(y_pack)) return LSTC_E_ALIGN;
    const int64_t chunks = rows * (int64_t)d / 8;
    hipLaunchKernelGGL(dropout_apply_pack_kernel, grid_for(chunks, NT * 2), NT, 0, (hipStream_t)stream, (const __bf16*)x_pack,
                       (__bf16*)y_pack, chunks, d / 32, d, make_drop_key(p, seed));
    return lstc_launch_status();
}

int lstc_splitk_finish(const float* parts, int32_t splits, int64_t part_stride, int64_t M, int64_t N, const float* bias,
                       const float* residual, int64_t ldr, const float* relu_src, int64_t ld_relu, float* C, int64_t ldc, int32_t flags,
                       float dropout_p, uint64_t dropout_seed, int32_t groups, int64_t group_stride_parts, int64_t group_stride_c,
                       void* stream) {
    if (!parts || !C) return LSTC_E_NULL;
    if (((flags & LSTC_EPI_BIAS) && !bias) || ((flags & LSTC_EPI_RESIDUAL) && !residual) || ((flags & LSTC_EPI_RELU_MASK) && !relu_src))
        return LSTC_E_NULL;
    if (splits <= 0 || M <= 0 || N <= 0 || part_stride < M * N || ldc < N || ((flags & LSTC_EPI_RESIDUAL) && ldr < N) ||
        ((flags & LSTC_EPI_RELU_MASK) && ld_relu < N) || !(dropout_p >= 0.f && dropout_p < 1.f)) return LSTC_E_SHAPE;
    if (flags & ~(LSTC_EPI_BIAS | LSTC_EPI_RELU | LSTC_EPI_DROPOUT | LSTC_EPI_RESIDUAL | LSTC_EPI_RELU_MASK | LSTC_EPI_ACCUM))
        return LSTC_E_UNSUPPORTED;
    if (groups < 1 || (groups > 1 && (group_stride_parts < (int64_t)splits * part_stride || group_stride_c <= 0))) return LSTC_E_SHAPE;
    if (groups > 1 && (flags & ~LSTC_EPI_ACCUM)) return LSTC_E_UNSUPPORTED;          // per-head groups: plain sums only
    if (M > 0x7fffffffLL || N > 0x7fffffffLL || (uint64_t)M * (uint64_t)N * (uint64_t)groups > 0xffffffffull) return LSTC_E_RANGE;
    if (group_stride_parts % 4 || group_stride_c % 4) return LSTC_E_ALIGN;
    if (N % 4 || part_stride % 4 || ldc % 4 || ((flags & LSTC_EPI_RESIDUAL) && ldr % 4) || ((flags & LSTC_EPI_RELU_MASK) && ld_relu % 4) ||
        !aligned16(parts) || !aligned16(C) || !aligned16(bias) || !aligned16(residual) || !aligned16(relu_src)) return LSTC_E_ALIGN;
    const DropKey dk = make_drop_key((flags & LSTC_EPI_DROPOUT) ? dropout_p : 0.f, dropout_seed);
    hipLaunchKernelGGL(splitk_finish_kernel, grid_for(M * (N / 4) * groups, NT), NT, 0, (hipStream_t)stream, parts, (int)splits, part_stride,
                       (int)M, (int)N, bias, residual, ldr, relu_src, ld_relu, C, ldc, (int)flags, dk, (int)groups, group_stride_parts,
                       group_stride_c);
    return lstc_launch_status();
}

int lstc_unpack1_rows(const void* x_pack, int64_t rows, int32_t K, int64_t row0, int64_t row_step, int64_t n, float* out, int64_t ldo,
                      void* stream) {
    if (!x_pack || !out) return LSTC_E_NULL;
    if (rows <= 0 || K <= 0 || n <= 0 || row0 < 0 || row_step <= 0 || row0 + (n - 1) * row_step >= rows || ldo < K) return LSTC_E_SHAPE;
    if (K % 8 != 0 || ldo % 4 != 0 || !aligned16(x_pack) || !aligned16(out)) return LSTC_E_ALIGN;
    const int kb = (K + 31) / 32, KBp = kb + (kb & 1);               // 32-k tiles per row block (even: csrc/gemm_bf16p.hip p1_kbp)
    hipLaunchKernelGGL(unpack1_rows_kernel, grid_for(n * (K / 8), NT), NT, 0, (hipStream_t)stream, (const __bf16*)x_pack, KBp, row0,
                       row_step, n, K, out, ldo);
    return lstc_launch_status();
}

int lstc_dropout_mask(uint8_t* mask, int64_t n, float p, uint64_t seed, void* stream) {
    if (!mask) return LSTC_E_NULL;
    if (n <= 0) return LSTC_E_SHAPE;
    if ((uint64_t)n > 0xffffffffull) return LSTC_E_RANGE;
    hipLaunchKernelGGL(dropout_mask_kernel, grid_for(n, NT * 4), NT, 0, (hipStream_t)stream, mask, n, make_drop_key(p, seed));
    return lstc_launch_status();
}

int lstc_head_out_fwd(const float* x, const float* W, const float* b, float* out, int64_t rows, int32_t c, void* stream) {
    if (!x || !W || !b || !out) return LSTC_E_NULL;
    if (rows <= 0 || (c != 1 && c != 2)) return LSTC_E_SHAPE;
    if (!aligned16(x)) return LSTC_E_ALIGN;
    hipLaunchKernelGGL(head_out_fwd_kernel, (unsigned)((rows + NT - 1) / NT), NT, 0, (hipStream_t)stream, x, W, b, out, rows, c);
    return lstc_launch_status();
}

int lstc_head_out_bwd(const float* x, const float* W, const float* out, const float* dout, float* dx, float* dW,
                      float* db, int64_t rows, int32_t c, void* stream) {
    if (!x || !W || !out || !dout || !dx || !dW || !db) return LSTC_E_NULL;
    if (rows <= 0 || (c != 1 && c != 2)) return LSTC_E_SHAPE;
    if (!aligned16(x) || !aligned16(dx)) return LSTC_E_ALIGN;
    hipLaunchKernelGGL(head_out_bwd_kernel, 1u, NT, 0, (hipStream_t)stream, x, W, out, dout, dx, dW, db, rows, c);
    return lstc_launch_status();
}

int lstc_adagrad_step(float* w, const float* grad, float* state, int64_t n, float lr, float weight_decay, float eps,
                      float gscale, void* stream) {
    if (!w || !grad || !state) return LSTC_E_NULL;
    if (n <= 0) return LSTC_E_SHAPE;
    hipLaunchKernelGGL(adagrad_kernel, grid_for(n, NT * 4), NT, 0, (hipStream_t)stream, w, grad, state, n, lr,
                       weight_decay, eps, gscale);
    return lstc_launch_status();
}

int lstc_adagrad_multi(const LstcAdagradItem* items, int32_t count, void* stream) {
    if (!items) return LSTC_E_NULL;
    if (count <= 0) return LSTC_E_SHAPE;
    for (int32_t i = 0; i < count; ++i) {
        if (!items[i].w || !items[i].grad || !items[i].state) return LSTC_E_NULL;
        if (items[i].n <= 0) return LSTC_E_SHAPE;
    }
    for (int32_t base = 0; base < count; base += ADA_MAX) {
        AdaBatch b;
        b.count = count - base < ADA_MAX ? count - base : ADA_MAX;
        int wg = 0;
        for (int i = 0; i < b.count; ++i) {
            const LstcAdagradItem& it = items[base + i];
            b.w[i] = it.w; b.g[i] = it.grad; b.s[i] = it.state; b.n[i] = it.n;
            b.lr[i] = it.lr; b.wd[i] = it.weight_decay; b.eps[i] = it.eps; b.gscale[i] = it.grad_scale;
            b.first_wg[i] = wg;
            const int64_t nwg = (it.n + ADA_PER_WG - 1) / ADA_PER_WG;
            if (nwg > 0x7fffffff - wg) return LSTC_E_RANGE;
            wg += (int)nwg;
        }
        b.first_wg[b.count] = wg;
        for (int i = b.count + 1; i <= ADA_MAX; ++i) b.first_wg[i] = wg;
        hipLaunchKernelGGL(adagrad_multi_kernel, dim3((unsigned)wg), NT, 0, (hipStream_t)stream, b);
        const int rc = lstc_launch_status();
        if (rc) return rc;
    }
    return 0;
}

static int vec_batch(const LstcVecItem* items, int32_t base, int32_t count, VecBatch& b) {
    b.count = count - base < ADA_MAX ? count - base : ADA_MAX;
    int wg = 0;
    for (int i = 0; i < b.count; ++i) {
        b.x[i] = items[base + i].x; b.n[i] = items[base + i].n;
        b.first_wg[i] = wg;
        const int64_t nwg = (items[base + i].n + ADA_PER_WG - 1) / ADA_PER_WG;
        if (nwg > 0x7fffffff - wg) return -1;
        wg += (int)nwg;
    }
    for (int i = b.count; i <= ADA_MAX; ++i) b.first_wg[i] = wg;
    return wg;
}

int64_t lstc_sqnorm_multi_scratch(const LstcVecItem* items, int32_t count) {
    if (!items || count <= 0) return 0;
    int64_t wg = 0;
    for (int32_t i = 0; i < count; ++i) {
        if (items[i].n <= 0) return 0;
        wg += (items[i].n + ADA_PER_WG - 1) / ADA_PER_WG;
    }
    return wg;
}

int lstc_sqnorm_multi(const LstcVecItem* items, int32_t count, float* scratch, int64_t scratch_floats, float* out, void* stream) {
    if (!items || !scratch || !out) return LSTC_E_NULL;
    if (count <= 0) return LSTC_E_SHAPE;
    for (int32_t i = 0; i < count; ++i) {
        if (!items[i].x) return LSTC_E_NULL;
        if (items[i].n <= 0) return LSTC_E_SHAPE;
    }
    const int64_t need = lstc_sqnorm_multi_scratch(items, count);
    if (scratch_floats < need) return LSTC_E_SHAPE;
    int64_t off = 0;
    for (int32_t base = 0; base < count; base += ADA_MAX) {
        VecBatch b;
        const int wg = vec_batch(items, base, count, b);
        if (wg < 0) return LSTC_E_RANGE;
        hipLaunchKernelGGL(sqnorm_multi_kernel, dim3((unsigned)wg), NT, 0, (hipStream_t)stream, b, scratch + off);
        const int rc = lstc_launch_status();
        if (rc) return rc;
        off += wg;
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), NT, 0, (hipStream_t)stream, scratch, need, out);
    return lstc_launch_status();
}

int lstc_clip_scale_multi(const LstcVecItem* items, int32_t count, const float* sqnorm, float max_norm, void* stream) {
    if (!items || !sqnorm) return LSTC_E_NULL;
    if (count <= 0 || !(max_norm > 0.f)) return LSTC_E_SHAPE;
    for (int32_t i = 0; i < count; ++i) {
        if (!items[i].x) return LSTC_E_NULL;
        if (items[i].n <= 0) return LSTC_E_SHAPE;
    }
    for (int32_t base = 0; base < count; base += ADA_MAX) {
        VecBatch b;
        const int wg = vec_batch(items, base, count, b);
        if (wg < 0) return LSTC_E_RANGE;
        hipLaunchKernelGGL(clip_scale_multi_kernel, dim3((unsigned)wg), NT, 0, (hipStream_t)stream, b, sqnorm, max_norm);
        const int rc = lstc_launch_status();
        if (rc) return rc;
    }
    return 0;
}

int lstc_scale(float* x, int64_t n, float alpha, void* stream) {
    if (!x) return LSTC_E_NULL;
    if (n <= 0) return LSTC_E_SHAPE;
    hipLaunchKernelGGL(scale_kernel, grid_for(n, NT * 4), NT, 0, (hipStream_t)stream, x, n, alpha);
    return lstc_launch_status();
}

int lstc_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream) {
    if (!x || !y) return LSTC_E_NULL;
    if (n <= 0) return LSTC_E_SHAPE;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, grid_for(n, NT * 4), NT, 0, (hipStream_t)stream, x, (__bf16*)y, n);
    return lstc_launch_status();
}

int lstc_cast_bf16_f32(const void* x, float* y, int64_t n, void* stream) {
    if (!x || !y) return LSTC_E_NULL;
    if (n <= 0) return LSTC_E_SHAPE;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, grid_for(n, NT * 4), NT, 0, (hipStream_t)stream, (const __bf16*)x, y, n);
    return lstc_launch_status();
}

int lstc_gather_rows(const float* src, int64_t src_rows, const int64_t* idx, float* dst, int64_t n_rows,
                     int64_t row_floats, void* stream) {
    if (!src || !idx || !dst) return LSTC_E_NULL;
    if (src_rows <= 0 || n_rows <= 0 || row_floats <= 0 || (row_floats & 3) || n_rows > 0x7fffffffLL) return LSTC_E_SHAPE;
    if (!aligned16(src) || !aligned16(dst)) return LSTC_E_ALIGN;
    const int64_t row_vec = row_floats / 4;
    int chunks = (int)((row_vec + NT * 4 - 1) / (NT * 4));        // ~4 float4 per thread
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_rows, (unsigned)chunks), NT, 0, (hipStream_t)stream,
                       (const float4*)src, src_rows, idx, (float4*)dst, row_vec);
    return lstc_launch_status();
}

int lstc_sqnorm_accum(const float* x, int64_t n, float* out, void* stream) {
    if (!x || !out) return LSTC_E_NULL;
    if (n <= 0) return LSTC_E_SHAPE;
    hipLaunchKernelGGL(sqnorm_kernel, grid_for(n, NT * 8, 1024), NT, 0, (hipStream_t)stream, x, n, out);
    return lstc_launch_status();
}

}  // extern "C"
