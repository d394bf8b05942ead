"""ctypes binding of liblstc_hip.so (the C ABI declared in include/lstc_hip.h).

The product path has NO fallback: if the shared library is missing or a tensor is not on a HIP
device, the call raises.  PyTorch is used only as plumbing here: it owns device memory and the
current HIP stream; every pointer handed to the library is ``tensor.data_ptr()``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first so its libamdhip64.so.7 is the one the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LSTC_LIBRARY: another build of the same library (tools/build_variant.sh -> build/<name>/liblstc_hip.so) for same-box A/B timings
LIB_PATH = os.environ.get("LSTC_LIBRARY") or os.path.join(_HERE, "liblstc_hip.so")

F32, BF16, F32X3, BF16P = 0, 1, 2, 3
EPI_BIAS, EPI_RELU, EPI_DROPOUT, EPI_RESIDUAL, EPI_RELU_MASK, EPI_ACCUM, EPI_OUT_F32 = 1, 2, 4, 8, 16, 32, 64
EPI_OUT_PACK, EPI_RELU_MASK_PACK, EPI_RESIDUAL_PACK = 128, 256, 512

EXPORTS = (
    "lstc_gemm", "lstc_attn_fwd", "lstc_attn_bwd", "lstc_attn_cls_fwd", "lstc_attn_cls_bwd", "lstc_cls_dot", "lstc_cls_wsum",
    "lstc_cls_outer", "lstc_cls_dot_pack", "lstc_cls_wsum_pack", "lstc_cls_outer_pack", "lstc_unpack1_rows", "lstc_splitk_finish", "lstc_layernorm_fwd", "lstc_layernorm_bwd", "lstc_layernorm_fwd_pack",
    "lstc_layernorm_bwd_drop_pack", "lstc_layernorm_bwd_drop", "lstc_layernorm_fwd_act", "lstc_layernorm_bwd_act",
    "lstc_cls_concat_fwd", "lstc_cls_concat_fwd_pack", "lstc_cls_concat_gather_fwd", "lstc_cls_concat_bwd", "lstc_colsum", "lstc_colsum_batched", "lstc_dropout_apply", "lstc_dropout_apply_pack", "lstc_dropout_mask", "lstc_dropout_seed_device",
    "lstc_head_out_fwd", "lstc_head_out_bwd", "lstc_vad_loss", "lstc_adagrad_step", "lstc_adagrad_multi", "lstc_sqnorm_accum", "lstc_scale",
    "lstc_sqnorm_multi_scratch", "lstc_sqnorm_multi", "lstc_clip_scale_multi",
    "lstc_gather_rows", "lstc_cast_f32_bf16", "lstc_cast_bf16_f32", "lstc_pack3", "lstc_pack3_bytes", "lstc_pack1", "lstc_pack1_multi", "lstc_pack1_bytes", "lstc_colsum_pack1", "lstc_gemm_splits",
    "lstc_version", "lstc_strerror",
)


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("transA", C.c_int32), ("transB", C.c_int32),
                ("dtype", C.c_int32), ("flags", C.c_int32),
                ("alpha", C.c_float), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
                ("ldr", C.c_int32), ("ld_relu", C.c_int32), ("split_k", C.c_int32), ("variant", C.c_int32),
                ("batch", C.c_int32), ("batch_stride_a", C.c_int64), ("batch_stride_b", C.c_int64), ("batch_stride_c", C.c_int64),
                ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("bias", C.c_void_p), ("residual", C.c_void_p), ("relu_src", C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("S", C.c_int32), ("H", C.c_int32), ("dk", C.c_int32), ("dv", C.c_int32),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
                ("dtype", C.c_int32), ("index_ld", C.c_int32), ("table_rows", C.c_int32),
                ("scale", C.c_float), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
                ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
                ("probs", C.c_void_p), ("table", C.c_void_p), ("index", C.c_void_p),
                ("dO", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
                ("dtable", C.c_void_p), ("dtable_chunks", C.c_int32), ("variant", C.c_int32),
                ("dQ_pack", C.c_void_p), ("dK_pack", C.c_void_p), ("dV_pack", C.c_void_p),
                ("pack_cols", C.c_int32), ("dQ_col0", C.c_int32), ("dK_col0", C.c_int32), ("dV_col0", C.c_int32),
                ("O_pack", C.c_void_p),
                ("in_pack_cols", C.c_int32), ("Q_col0", C.c_int32), ("K_col0", C.c_int32), ("V_col0", C.c_int32),
                ("dO_pack_cols", C.c_int32), ("dO_col0", C.c_int32), ("probs_ld", C.c_int32)]


class PackItem(C.Structure):
    """LstcPackItem (include/lstc_hip.h)."""
    _fields_ = [("src", C.c_void_p), ("rows", C.c_int64), ("K", C.c_int64), ("ld", C.c_int64), ("k_major", C.c_int32), ("dst", C.c_void_p)]


class AdagradItem(C.Structure):
    """LstcAdagradItem (include/lstc_hip.h)."""
    _fields_ = [("w", C.c_void_p), ("grad", C.c_void_p), ("state", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_float), ("weight_decay", C.c_float), ("eps", C.c_float), ("grad_scale", C.c_float)]


class VecItem(C.Structure):
    """LstcVecItem (include/lstc_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("n", C.c_int64)]


class LossDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("bs_global", C.c_int32), ("bs_local", C.c_int32), ("rank_off", C.c_int32),
                ("part_num", C.c_int32), ("score_len", C.c_int32), ("label_len", C.c_int32), ("l1_skip", C.c_int32),
                ("lambda_1", C.c_float), ("lambda_MIL", C.c_float), ("lambda_aux", C.c_float),
                ("lambda_normal", C.c_float), ("lambda_abnormal", C.c_float),
                ("out", C.c_void_p), ("abn_labels", C.c_void_p), ("targets", C.c_void_p), ("bag", C.c_void_p), ("dout", C.c_void_p),
                ("scalars", C.c_void_p), ("phase", C.c_int32)]


_lib = None


def load():
    """Load the library once; raise loudly when it is absent (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"lstc_vad_amd: {LIB_PATH} not found. Build it with `make` (or `python -c 'import __graft_entry__ as g; "
            "g.build()'`). There is no fallback path: the HIP library is required.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64
    sig = {
        "lstc_gemm": [C.POINTER(GemmDesc), vp],
        "lstc_attn_fwd": [C.POINTER(AttnDesc), vp],
        "lstc_attn_bwd": [C.POINTER(AttnDesc), vp],
        "lstc_attn_cls_fwd": [C.POINTER(AttnDesc), vp],
        "lstc_attn_cls_bwd": [C.POINTER(AttnDesc), vp],
        "lstc_cls_dot": [vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, u64, vp],
        "lstc_cls_wsum": [vp, vp, vp, i64, i32, i32, i32, vp],
        "lstc_cls_outer": [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp],
        "lstc_cls_dot_pack": [vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, u64, vp],
        "lstc_cls_wsum_pack": [vp, vp, vp, i64, i32, i32, i32, vp],
        "lstc_cls_outer_pack": [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp],
        "lstc_unpack1_rows": [vp, i64, i32, i64, i64, i64, vp, i64, vp],
        "lstc_splitk_finish": [vp, i32, i64, i64, i64, vp, vp, i64, vp, i64, vp, i64, i32, f32, u64, i32, i64, i64, vp],
        "lstc_layernorm_fwd": [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp],
        "lstc_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp],
        "lstc_layernorm_fwd_pack": [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp, vp],
        "lstc_layernorm_bwd_drop_pack": [vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, f32, u64, vp, vp],
        "lstc_layernorm_bwd_drop": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, f32, u64, vp],
        "lstc_layernorm_fwd_act": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp],
        "lstc_layernorm_bwd_act": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, f32, u64, vp, vp],
        "lstc_cls_concat_fwd": [vp, vp, i64, vp, vp, vp, i64, i32, i32, vp],
        "lstc_cls_concat_fwd_pack": [vp, vp, i64, vp, vp, vp, i64, i32, i32, vp, vp],
        "lstc_cls_concat_gather_fwd": [vp, i64, vp, i32, vp, vp, vp, i64, i32, i32, vp, vp],
        "lstc_cls_concat_bwd": [vp, vp, i64, i32, i32, i32, vp],
        "lstc_colsum": [vp, i64, i32, i32, vp, i32, vp, i32, vp],
        "lstc_colsum_batched": [vp, i32, i64, i32, i32, i64, vp, i32, vp, vp],
        "lstc_dropout_apply": [vp, vp, i64, f32, u64, vp],
        "lstc_dropout_apply_pack": [vp, vp, i64, i32, f32, u64, vp],
        "lstc_dropout_mask": [vp, i64, f32, u64, vp],
        "lstc_dropout_seed_device": [vp],
        "lstc_head_out_fwd": [vp, vp, vp, vp, i64, i32, vp],
        "lstc_head_out_bwd": [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp],
        "lstc_vad_loss": [C.POINTER(LossDesc), vp],
        "lstc_adagrad_step": [vp, vp, vp, i64, f32, f32, f32, f32, vp],
        "lstc_adagrad_multi": [C.POINTER(AdagradItem), i32, vp],
        "lstc_sqnorm_accum": [vp, i64, vp, vp],
        "lstc_scale": [vp, i64, f32, vp],
        "lstc_sqnorm_multi_scratch": [C.POINTER(VecItem), i32],
        "lstc_sqnorm_multi": [C.POINTER(VecItem), i32, vp, i64, vp, vp],
        "lstc_clip_scale_multi": [C.POINTER(VecItem), i32, vp, f32, vp],
        "lstc_gather_rows": [vp, i64, vp, vp, i64, i64, vp],
        "lstc_cast_f32_bf16": [vp, vp, i64, vp],
        "lstc_cast_bf16_f32": [vp, vp, i64, vp],
        "lstc_pack3": [vp, i64, i64, i64, C.c_int32, vp, vp],
        "lstc_pack3_bytes": [i64, i64],
        "lstc_pack1": [vp, i64, i64, i64, C.c_int32, vp, vp],
        "lstc_pack1_multi": [vp, C.c_int32, vp],
        "lstc_colsum_pack1": [vp, i64, i32, vp, i32, vp, i32, vp],
        "lstc_pack1_bytes": [i64, i64],
        "lstc_gemm_splits": [i32, i32, i32],
        "lstc_version": [],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.lstc_pack3_bytes.restype = C.c_int64
    lib.lstc_pack1_bytes.restype = C.c_int64
    lib.lstc_sqnorm_multi_scratch.restype = C.c_int64
    lib.lstc_strerror.argtypes = [C.c_int]
    lib.lstc_strerror.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().lstc_strerror(rc).decode()
        raise RuntimeError(f"{what or 'lstc'} failed: {msg} (code {rc})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dev_ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: the product has no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("lstc_vad_amd: tensor is not on a HIP device; the hot path is HIP-only (no CPU fallback)")
    return t.data_ptr()
