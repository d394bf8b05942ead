// C-ABI front door: dtype dispatch, version, error strings.
#include "lstc_common.h"

__attribute__((visibility("hidden"))) int lstc_gemm_f32_impl(const LstcGemmDesc* d, hipStream_t st);
__attribute__((visibility("hidden"))) int lstc_gemm_bf16_impl(const LstcGemmDesc* d, hipStream_t st) __attribute__((weak));
__attribute__((visibility("hidden"))) int lstc_gemm_f32x3_impl(const LstcGemmDesc* d, hipStream_t st);
__attribute__((visibility("hidden"))) int lstc_gemm_bf16p_impl(const LstcGemmDesc* d, hipStream_t st);

static std::atomic<const uint64_t*> g_seed_dev{nullptr};
const uint64_t* lstc_seed_dev_current() { return g_seed_dev.load(std::memory_order_acquire); }

extern "C" {

int lstc_dropout_seed_device(const uint64_t* dev_word) {
    g_seed_dev.store(dev_word, std::memory_order_release);
    return LSTC_OK;
}

int lstc_gemm(const LstcGemmDesc* d, void* stream) {
    if (!d) return LSTC_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == LSTC_F32) return lstc_gemm_f32_impl(d, st);
    if (d->dtype == LSTC_BF16 && lstc_gemm_bf16_impl) return lstc_gemm_bf16_impl(d, st);
    if (d->dtype == LSTC_F32X3) return lstc_gemm_f32x3_impl(d, st);
    if ((d->flags & (LSTC_EPI_OUT_PACK | LSTC_EPI_RELU_MASK_PACK | LSTC_EPI_RESIDUAL_PACK)) && d->dtype != LSTC_BF16P) return LSTC_E_UNSUPPORTED;
    if (d->dtype == LSTC_BF16P) return lstc_gemm_bf16p_impl(d, st);
    return LSTC_E_UNSUPPORTED;
}

int32_t lstc_gemm_splits(int32_t dtype, int32_t K, int32_t split_k) {
    if (K <= 0) return 0;
    const int bk = (dtype == LSTC_BF16 || dtype == LSTC_BF16P) ? 64 : 32;   // K tile of gemm_bf16c, gemm_bf16p / gemm_f32, gemm_pk
    const int s = split_k > 1 ? split_k : 1;
    const int kt = (K + bk - 1) / bk;
    const int per = (kt + s - 1) / s;
    return (kt + per - 1) / per;
}

int lstc_version(void) { return LSTC_VERSION; }

const char* lstc_strerror(int code) {
    switch (code) {
        case LSTC_OK: return "ok";
        case LSTC_E_NULL: return "lstc: required pointer is NULL";
        case LSTC_E_SHAPE: return "lstc: bad or inconsistent dimension";
        case LSTC_E_ALIGN: return "lstc: pointer or leading dimension not aligned as required";
        case LSTC_E_UNSUPPORTED: return "lstc: unsupported combination";
        case LSTC_E_RANGE: return "lstc: size exceeds a documented limit";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "lstc: unknown error";
}

}  // extern "C"
