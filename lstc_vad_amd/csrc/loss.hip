// MIL ranking / sparsity / CE / BCE loss and its gradient in ONE launch (gfx950).
// The reference evaluates the hinge with a Python loop of `bs` tiny kernels
// (Train/temporal_transformer_shanghaitech.py:29-31) and formats four device scalars per step (:143);
// here one 256-thread workgroup does everything: the score vector is at most a few thousand floats, so
// the kernel is latency-bound by construction and the design goal is simply "one launch, no host sync".
#include "lstc_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXV = 4096;   // 2 * bs_global limit (LDS)

struct LossParams {
    LstcLossDesc d;
};

__device__ __forceinline__ float score_at(const LstcLossDesc& d, int c, int e) { return d.out[(size_t)e * c + (c - 1)]; }

__global__ void __launch_bounds__(NT) vad_loss_kernel(const LstcLossDesc d) {
    __shared__ float bag[MAXV];          // global bag vector (normal half, abnormal half)
    __shared__ float gbag[MAXV];         // d loss / d bag for this rank's videos (local numbering)
    __shared__ int argp[MAXV];           // argmax part of this rank's videos
    __shared__ float red[NT / 64];
    const int c = d.mode == 1 ? 2 : 1;
    const int Ls = d.score_len, pn = d.part_num;
    const int rpv = pn * Ls;                         // scores per video
    const int nloc = 2 * d.bs_local;                 // local videos
    const int bsg = d.bs_global, bsl = d.bs_local;
    const int nelem = nloc * rpv;                    // local score elements

    // ---- 1. bag maxima + argmax of this rank's videos
    for (int v = threadIdx.x; v < nloc; v += NT) {
        float best = -INFINITY;
        int bp = 0;
        for (int p = 0; p < pn; ++p) {
            float s = 0.f;
            for (int l = 0; l < Ls; ++l) s += score_at(d, c, (v * pn + p) * Ls + l);
            s = s / (float)Ls;
            if (s > best) { best = s; bp = p; }
        }
        argp[v] = bp;
        gbag[v] = best;       // temporarily holds the bag value
    }
    __syncthreads();
    if (d.phase == 0) {
        for (int v = threadIdx.x; v < nloc; v += NT) {
            const int g = v < bsl ? d.rank_off + v : bsg + d.rank_off + (v - bsl);
            d.bag[g] = gbag[v];
        }
        return;
    }
    // ---- 2. global bag vector into LDS
    if (d.phase == 2) {
        for (int v = threadIdx.x; v < nloc; v += NT) bag[v] = gbag[v];      // single rank: local == global
        if (d.bag)
            for (int v = threadIdx.x; v < nloc; v += NT) d.bag[v] = gbag[v];
    } else {
        for (int v = threadIdx.x; v < 2 * bsg; v += NT) bag[v] = d.bag[v];
    }
    __syncthreads();
    // ---- 3. hinge: err = sum_i sum_j relu(1 - abn_j + nor_i) / bs^2 ; pair (i, j) is booked on the rank owning i
    const float inv_b2 = 1.f / ((float)bsg * (float)bsg);
    float err = 0.f;
    for (int v = threadIdx.x; v < nloc; v += NT) {
        float g = 0.f;
        if (v < bsl) {
            const float nor = bag[d.rank_off + v];
            float cnt = 0.f;
            for (int j = 0; j < bsg; ++j) {
                const float t = 1.f - bag[bsg + j] + nor;
                if (t > 0.f) { cnt += 1.f; err += t; }
            }
            g = cnt * inv_b2;
        } else {
            const float abn = bag[bsg + d.rank_off + (v - bsl)];
            float cnt = 0.f;
            for (int i = 0; i < bsg; ++i)
                if (1.f - abn + bag[i] > 0.f) cnt += 1.f;
            g = -cnt * inv_b2;
        }
        gbag[v] = g;
    }
    err = block_sum<NT>(err, red) * inv_b2;
    __syncthreads();
    // ---- 4. per-element pass: gradient + l1 / aux partial sums
    const float n_l1 = (float)(2 * bsg * rpv - d.l1_skip);
    const bool has_aux = (d.abn_labels != nullptr || d.targets != nullptr) && d.mode != 0;
    float l1 = 0.f, aux = 0.f;
    const float n_rows_g = (float)(2 * bsg * rpv);          // CE: mean over all global rows (mode 1: rpv = pn)
    const float n_parts_g = (float)(2 * bsg * pn);          // BCE: mean over [2bs, pn]
    for (int e = threadIdx.x; e < nelem; e += NT) {
        const int v = e / rpv, rem = e % rpv, part = rem / Ls;
        const int ge = v < bsl ? d.rank_off * rpv + e : bsg * rpv + d.rank_off * rpv + (e - bsl * rpv);
        const float sc = score_at(d, c, e);
        float g = 0.f;
        if (part == argp[v]) g += gbag[v] / (float)Ls;
        if (ge >= d.l1_skip) {
            l1 += sc;
            g += d.lambda_1 / n_l1;
        }
        g *= d.lambda_MIL;
        // soft target of this part: normal -> t1 = 0; abnormal -> mean of its pseudo labels
        float t1 = 0.f, t0 = 1.f;
        if (has_aux && d.targets) {
            t0 = d.targets[(size_t)(v * pn + part) * 2];
            t1 = d.targets[(size_t)(v * pn + part) * 2 + 1];
        } else if (has_aux && v >= bsl) {
            const float* lab = d.abn_labels + ((size_t)(v - bsl) * pn + part) * d.label_len;
            float s = 0.f;
            for (int l = 0; l < d.label_len; ++l) s += lab[l];
            t1 = s / (float)d.label_len;
            t0 = 1.f - t1;
        }
        if (d.mode == 1) {
            float g0 = 0.f;
            if (has_aux) {
                const float p0 = d.out[(size_t)e * 2], p1 = sc;
                const float m = fmaxf(p0, p1);
                const float lse = m + logf(expf(p0 - m) + expf(p1 - m));
                const float ls0 = p0 - lse, ls1 = p1 - lse;
                aux += -(t0 * ls0 + t1 * ls1);
                const float tsum = t0 + t1;
                g0 = d.lambda_aux * (tsum * expf(ls0) - t0) / n_rows_g;
                g += d.lambda_aux * (tsum * expf(ls1) - t1) / n_rows_g;
            }
            d.dout[(size_t)e * 2] = g0;
            d.dout[(size_t)e * 2 + 1] = g;
        } else {
            if (d.mode == 2 && has_aux) {
                float o = 0.f;
                for (int l = 0; l < Ls; ++l) o += score_at(d, c, (v * pn + part) * Ls + l);
                o = o / (float)Ls;
                const float a = 1.f - o + 1e-8f, b = o + 1e-8f;
                if (rem % Ls == 0) aux += -d.lambda_normal * t0 * logf(a) - d.lambda_abnormal * t1 * logf(b);
                g += d.lambda_aux * (d.lambda_normal * t0 / a - d.lambda_abnormal * t1 / b) / n_parts_g / (float)Ls;
            }
            d.dout[e] = g;
        }
    }
    l1 = block_sum<NT>(l1, red) / n_l1;
    aux = block_sum<NT>(aux, red) / (d.mode == 1 ? n_rows_g : n_parts_g);
    if (threadIdx.x == 0) {
        const float mil = err + d.lambda_1 * l1;
        d.scalars[0] = d.lambda_MIL * mil + d.lambda_aux * aux;
        d.scalars[1] = mil;
        d.scalars[2] = err;
        d.scalars[3] = l1;
        d.scalars[4] = aux;
    }
}

}  // namespace

extern "C" int lstc_vad_loss(const LstcLossDesc* d, void* stream) {
    if (!d) return LSTC_E_NULL;
    if (!d->out) return LSTC_E_NULL;
    if (d->phase == 0 && !d->bag) return LSTC_E_NULL;
    if (d->phase >= 1 && (!d->dout || !d->scalars)) return LSTC_E_NULL;
    if (d->phase == 1 && !d->bag) return LSTC_E_NULL;
    if (d->mode < 0 || d->mode > 2 || d->phase < 0 || d->phase > 2) return LSTC_E_UNSUPPORTED;
    if (d->bs_global <= 0 || d->bs_local <= 0 || d->part_num <= 0 || d->score_len <= 0 || d->rank_off < 0) return LSTC_E_SHAPE;
    if (d->rank_off + d->bs_local > d->bs_global) return LSTC_E_SHAPE;
    if (d->phase == 2 && d->bs_local != d->bs_global) return LSTC_E_SHAPE;
    if (d->mode == 1 && d->score_len != 1) return LSTC_E_SHAPE;
    if (d->abn_labels && !d->targets && d->label_len <= 0) return LSTC_E_SHAPE;
    if (2 * d->bs_global > MAXV) return LSTC_E_RANGE;
    if (d->l1_skip < 0 || d->l1_skip >= 2 * d->bs_global * d->part_num * d->score_len) return LSTC_E_SHAPE;
    hipLaunchKernelGGL(vad_loss_kernel, 1, NT, 0, (hipStream_t)stream, *d);
    return lstc_launch_status();
}
