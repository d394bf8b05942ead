"""Thin op layer over the C ABI plus the autograd Functions the model classes use.

Every function here launches hand-written HIP kernels from liblstc_hip.so on the current stream;
nothing falls back to ATen compute.  torch supplies allocation (``torch.empty``), views and the
autograd graph.  Shapes follow the reference: tokens are kept as ``[M = N*S, d]``
row-major matrices from the CLS concat to the head, so no transposes or copies appear between ops.
"""
from __future__ import annotations

import ctypes as C
import math
import functools
import os
import weakref
from contextlib import contextmanager
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_ACCUM, EPI_BIAS, EPI_DROPOUT, EPI_OUT_PACK, EPI_RELU, EPI_RELU_MASK, EPI_RELU_MASK_PACK, EPI_RESIDUAL, EPI_RESIDUAL_PACK, F32, AttnDesc, GemmDesc,
                   LossDesc, check, dev_ptr, stream_ptr)

# ------------------------------------------------------------------------------------------ RNG
_MASK64 = (1 << 64) - 1
_counter = 0
_recorder = None          # list collecting (site, p, seed, shape) while record_dropout() is active


def next_seed() -> int:
    """64-bit seed for one dropout site invocation: a hash of torch's seed PLUS a call counter, so a run is reproducible
    under ``torch.manual_seed`` without touching torch's generators.  Linear in the counter on purpose: a captured step
    (engine.GraphedStep) replays with  seed + device word  (lstc_dropout_seed_device) and must land on the seeds the eager
    steps draw; the kernels' key derivation and per-element hash (csrc/lstc_common.h) do the mixing."""
    global _counter
    _counter += 1
    x = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0xD1342543DE82EF95) & _MASK64
    x ^= x >> 32
    x = (x * 0xD6E8FEB86659FD93) & _MASK64
    x ^= x >> 32
    return (x + _counter) & _MASK64


def reset_rng(counter: int = 0):
    global _counter
    _counter = counter


@contextmanager
def record_dropout():
    """Collect every dropout site fired inside the block as ``(site, p, seed, shape)`` so a test can
    regenerate the masks (``dropout_mask``) and replay the step through the oracle."""
    global _recorder
    prev, _recorder = _recorder, []
    try:
        yield _recorder
    finally:
        _recorder = prev


def _note(site, p, seed, shape):
    if _recorder is not None:
        _recorder.append((site, float(p), int(seed), tuple(shape)))


_gemm_prof = None         # list of (flops, start_event, end_event) while bench.py profiles GEMM launches
_compute_dtype = F32      # LstcGemmDesc.dtype used by every GEMM: F32 = exact f32 MFMA, BF16 = bf16 MFMA on f32 storage


def set_compute_dtype(name: str):
    """"fp32" (default; exact-f32 MFMA, the parity mode), "f32x3" (f32-accurate products on the 16-bit matrix cores: every
    operand scaled by a power of two and split into two f16 planes, three plane products, f32 accumulation -
    csrc/gemm_pk.hip; small GEMMs stay on the exact-f32 kernel) or "bf16" (operands rounded to bf16 RNE, f32 accumulate and
    f32 storage everywhere: BASELINE.json configs 3 / 5; large products run on packed bf16 tiles - lstc_pack1 +
    csrc/gemm_bf16p.hip - small or batched ones convert while staging, csrc/gemm_bf16c.hip; the attention products Q K^T, P V
    and their gradients round their operands to bf16 the same way, LstcAttnDesc.dtype = LSTC_BF16, unless LSTC_ATTN_F32=1).
    Softmax, LayerNorm, loss and Adagrad stay f32."""
    global _compute_dtype
    if name in ("fp32", "f32", "float32"):
        _compute_dtype = F32
    elif name in ("bf16", "bfloat16"):
        _compute_dtype = _lib.BF16
    elif name in ("f32x3", "fp32x3"):
        _compute_dtype = _lib.F32X3
    else:
        raise ValueError(name)


def get_compute_dtype() -> str:
    return {_lib.BF16: "bf16", _lib.F32X3: "f32x3"}.get(_compute_dtype, "fp32")


# ---- bf16 activation stream (round 5).  bf16 mode only: between the CLS concat and the last full encoder layer every activation
# and every gradient of the residual stream lives ONLY as an lstc_pack1 operand (2 B per element): the GEMM epilogues read the
# residual from a pack and write the pre-LayerNorm sum as a pack (LSTC_EPI_RESIDUAL_PACK + LSTC_EPI_OUT_PACK), the LayerNorm kernels
# run on packs (lstc_layernorm_fwd_act / _bwd_act).  "fp32" keeps rounds 1-4's f32 activations between the blocks.
_ACT16 = os.environ.get("LSTC_ACT_DTYPE", "bf16").lower() not in ("fp32", "f32", "float32")


def set_act_dtype(name: str):
    """"bf16" (default) | "fp32": storage of the residual stream between encoder blocks in the bf16 compute mode."""
    global _ACT16
    if name in ("bf16", "bfloat16"):
        _ACT16 = True
    elif name in ("fp32", "f32", "float32"):
        _ACT16 = False
    else:
        raise ValueError(name)


def get_act_dtype() -> str:
    return "bf16" if (_ACT16 and _compute_dtype == _lib.BF16) else "fp32"


class PackedAct:
    """An activation [N, S, d] of the bf16 stream as the model classes pass it between blocks: ``t`` is the autograd-tracked
    torch.bfloat16 view of the lstc_pack1 buffer of the [N*S, d] matrix (its gradient is a buffer of the same layout), ``shape``
    the logical shape."""
    __slots__ = ("t", "shape")

    def __init__(self, t, shape):
        self.t, self.shape = t, tuple(shape)

    def pack(self) -> "Packed":
        return _act_pack(self.t, self.shape[0] * self.shape[1], self.shape[2])


def _act_pack(t: torch.Tensor, rows: int, d: int) -> "Packed":
    """The ``Packed`` operand behind a bf16 stream tensor."""
    if t.dtype != torch.bfloat16 or t.dim() != 1 or not t.is_contiguous() or \
            t.numel() * 2 != int(_lib.load().lstc_pack1_bytes(rows, d)):
        raise RuntimeError(f"bf16 activation stream: expected the bf16 view of an lstc_pack1 buffer of [{rows}, {d}], got "
                           f"{tuple(t.shape)} {t.dtype}")
    return Packed(t.view(torch.uint8), rows, d, _lib.BF16P)


def _act_tensor(pk: "Packed") -> torch.Tensor:
    return pk.buf.view(torch.bfloat16)


def act_rows_ok(rows: int, d: int) -> bool:
    """[rows, d] activations the bf16 stream can carry (include/lstc_hip.h, lstc_layernorm_fwd_act)."""
    return _ACT16 and d in (1024, 2048) and _fused_pack_shape(rows, d) and rows * d <= 0xffffffff


def act_chain_ok(N: int, S: int, dm: int, layers) -> bool:
    """Can the full encoder layers ``layers`` run on the bf16 activation stream for N sequences of S tokens?  Everything the act
    paths of MHAFunction / FFNFunction assume: LayerNorm after both blocks, fused Q|K|V weights, the packed-operand attention
    kernels, every product of the block on whole 256-tiles of the packed kernel."""
    M = N * S
    if not layers or not act_rows_ok(M, dm):
        return False
    for layer in layers:
        a, f = layer.slf_attn, layer.pos_ffn
        # the FFN needs its LayerNorm (the stream's f32 exit - where the CLS-only layer or the caller wants rows - is an LN kernel); the
        # attention block runs with or without one (STN configs: MHA_layerNorm = False, its sum dropout(fc(o)) + x IS the output)
        if not (layer.FFN_need and f.layerNorm_flag and a.d_model == dm):
            return False
        H, dk, dv = a.n_head, a.d_k, a.d_v
        wqkv = _fused_qkv_weight(a.w_qs.weight, a.w_ks.weight, a.w_vs.weight)
        if wqkv is None or wqkv.shape[0] != H * (2 * dk + dv):
            return False
        if not (attn_packed_inputs(N, S, H, dk, dv) and packed_out_shape(M, wqkv.shape[0]) and packed_out_shape(M, H * dv) and
                packed_out_shape(M, dm) and a.fc.weight.shape[0] >= max(_x3_min[0], 1)):
            return False
        Fp = _padded_hidden(f.w_1.weight.shape[0])
        if not (packed_out_shape(M, Fp) and f.w_2.weight.shape[0] >= max(_x3_min[0], 1)):
            return False
    return True


# ---- packed operands of the f32x3 GEMM (csrc/gemm_pk.hip) ------------------------------------------------------------
class Packed:
    """An operand as lstc_pack3 / lstc_pack1 leave it: logical [rows, K] in the GEMM's LDS-image tiling (``kind`` F32X3: two
    scaled f16 planes + scale trailer; BF16P: one bf16 plane)."""
    __slots__ = ("buf", "rows", "K", "kind")

    def __init__(self, buf, rows, K, kind):
        self.buf, self.rows, self.K, self.kind = buf, rows, K, kind


def _packed_kind():
    """GEMM dtype code of the packed-operand kernel of the current compute mode (None in exact-f32 mode)."""
    if _compute_dtype == _lib.F32X3:
        return _lib.F32X3
    if _compute_dtype == _lib.BF16:
        return _lib.BF16P
    return None


_pack_prof = None          # list of (bytes_in, start_event, end_event) while bench.py profiles
_wepoch = 0
_memo_stack = []           # activation packs made inside one autograd-node body are shared by the GEMMs of that body
_x3_min = (256, 256, 1 << 30)   # min(M, N), K, M*N*K from which a product goes to the packed kernel
_BF16P_MIN_MNK = int(os.environ.get("LSTC_BF16P_MIN_MNK", str(1 << 28)))      # the M*N*K bound of the packed bf16 kernel (A/B: 1073741824 = rounds 2 - 5)
_CLS_PACK = os.environ.get("LSTC_CLS_PACK", "1") != "0"             # A/B hook: 0 = the CLS-only layer reads f32 rows (round 5's first form)
_ATTN_F32 = os.environ.get("LSTC_ATTN_F32", "0") == "1"          # bf16 mode: keep the attention products on the exact-f32 MFMA
_ATTN_VARIANT = int(os.environ.get("LSTC_ATTN_VARIANT", "0"))     # 1: first-generation attention kernels (A/B measurements)
_BWD_NPW = int(os.environ.get("LSTC_ATTN_BWD_NPW", "0"))       # measurement hook: sequences per workgroup of the attention backward
_ATTN_PACKED_IN = os.environ.get("LSTC_ATTN_PACKED_IN", "1") != "0"   # bf16 mode: Q | K | V / dO reach the attention core as packs
_WGRAD_COST_MODEL = os.environ.get("LSTC_WGRAD_COST_MODEL", "0") == "1"     # opt-in (measured, round 6: profiles/r06_wgrad_split_ab.txt): the cost-model split of the
# packed bf16 weight gradients gains 3 % on the UCF / STN rank shapes (4352 - 4864 tokens) and nothing elsewhere; it stays off by default - fewer partial
# buffers move the peak-memory comparison of tests/test_act16_gpu.py (both activation dtypes lose the same transient, the f32-activation run more)
_DETERMINISTIC_WGRAD = os.environ.get("LSTC_ATOMIC_SPLITK", "0") != "1"   # split-K weight gradients: partials + ordered sum, not atomics


def set_x3_threshold(min_mn=256, min_k=256, min_mnk=1 << 30):
    """Size from which f32x3 mode uses the packed kernel (tests set 0, 0, 0 to push the reduced-width cases through it)."""
    global _x3_min
    _x3_min = (int(min_mn), int(min_k), int(min_mnk))


def bump_weight_epoch(params=None):
    """Invalidate packed-weight copies: the optimizer rewrote weights through raw pointers.  With ``params`` (what an optimizer
    step passes: the tensors IT updated) only those tensors' packs go stale - staleness is tracked per parameter
    (``_lstc_wver``), so two optimizers stepped back to back (engine.MixedStep, BASELINE config 5) do not invalidate each
    other's freshly rebuilt packs; without it every pack of the process does (graph replays, tests)."""
    global _wepoch
    if params is None:
        _wepoch += 1
    else:
        for p in params:
            p.__dict__["_lstc_wver"] = p.__dict__.get("_lstc_wver", 0) + 1
    _producer_packs.clear()


def _wstamp(t):
    """(process epoch, version of the weight - for a fused Q|K|V view: of the parameter it starts at)."""
    base = t.__dict__.get("_lstc_view_of")
    return (_wepoch, (base if base is not None else t).__dict__.get("_lstc_wver", 0))


# Packs that a row-wise kernel emitted next to its f32 result (lstc_layernorm_fwd_pack): the next block finds the pack of its
# input here instead of running lstc_pack1.  Entries hold the producing tensor (so its address cannot be recycled while the
# entry lives) and its version counter (an in-place write invalidates the pack); consumed on first use, two entries at most.
_producer_packs = {}
_FUSE_PACKS = os.environ.get("LSTC_NO_FUSED_PACKS", "0") != "1"


def _pack_key(t):
    return (t.data_ptr(), tuple(t.shape), t.stride())


def _register_pack(t: torch.Tensor, pk: "Packed"):
    while len(_producer_packs) >= 2:
        _producer_packs.pop(next(iter(_producer_packs)))
    _producer_packs[_pack_key(t)] = (t, t._version, pk)


def drop_producer_packs():
    """Forget the packs no GEMM picked up.  Encoder.forward / forward_cls call it on the way out, so an entry (it holds the
    producing f32 tensor and its pack: 1.2 GB at the headline shape) never outlives one encoder call - evaluation and
    pseudo-label loops never reach the optimizer's bump_weight_epoch()."""
    _producer_packs.clear()


def _producer_pack(t: torch.Tensor, kind):
    hit = _producer_packs.pop(_pack_key(t), None)
    if hit is None or hit[1] != t._version or hit[0]._version != hit[1] or hit[2].kind != kind:
        return None
    return hit[2]


def _fused_pack_shape(rows: int, d: int) -> bool:
    """A [rows, d] activation whose products go to the packed bf16 kernel and whose rows fill the pack's tile grid exactly
    (include/lstc_hip.h, lstc_layernorm_fwd_pack): the row-wise producer may emit its pack."""
    return (_FUSE_PACKS and _packed_kind() == _lib.BF16P and rows % 256 == 0 and d % 64 == 0 and d <= 2048 and
            rows >= max(_x3_min[0], 1) and d >= max(_x3_min[1], 256) and rows * d * max(_x3_min[0], 256) >= _x3_min[2])


class pack_memo:
    """``with pack_memo():`` - activation operands packed inside the block are reused by later GEMMs of the block (X feeds
    the Q, K and V projections; the k-major pack of X feeds three weight gradients)."""

    def __enter__(self):
        _memo_stack.append({})
        return self

    def __exit__(self, *exc):
        _memo_stack.pop()
        return False


def pack3(t: torch.Tensor, k_major: bool = False, kind=None) -> Packed:
    """Pack a 2-D f32 operand for the packed-operand GEMM of the current mode (f32x3: lstc_pack3, bf16: lstc_pack1).
    ``k_major`` = the contraction runs along dim 0 of ``t``."""
    kind = _packed_kind() if kind is None else kind
    pt, r, c, ld = _mat(t)
    rows, K = (c, r) if k_major else (r, c)
    lib = _lib.load()
    nbytes, fn, name = (lib.lstc_pack3_bytes, lib.lstc_pack3, "lstc_pack3") if kind == _lib.F32X3 else \
        (lib.lstc_pack1_bytes, lib.lstc_pack1, "lstc_pack1")
    buf = torch.empty((int(nbytes(rows, K)),), device=t.device, dtype=torch.uint8)
    if _pack_prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(fn(pt, rows, K, ld, int(k_major), dev_ptr(buf), stream_ptr()), name)
        e1.record()
        _pack_prof.append((4.0 * rows * K, e0, e1))
    else:
        check(fn(pt, rows, K, ld, int(k_major), dev_ptr(buf), stream_ptr()), name)
    return Packed(buf, rows, K, kind)


def _packed_operand(t, k_major):
    if isinstance(t, Packed):
        return t
    if (t.is_leaf and t.requires_grad) or getattr(t, "_lstc_weight_view", False):      # a weight: one pack per optimizer step and layout
        # cached ON the parameter object (a key built from data_ptr would outlive the tensor and hand a stale pack to the
        # next model that lands on the same address); autograd returns the same object from ctx.saved_tensors for leaves
        cache = t.__dict__.setdefault("_lstc_packs", {})
        ck = (k_major, _packed_kind())
        hit = cache.get(ck)
        if hit is not None and hit[0] == _wstamp(t) and hit[1] == t._version and hit[3] == t.data_ptr():
            return hit[2]
        pk = pack3(t.detach(), k_major)
        cache[ck] = (_wstamp(t), t._version, pk, t.data_ptr())
        return pk
    if not k_major and _producer_packs:
        hit = _producer_pack(t, _packed_kind())
        if hit is not None:
            if _memo_stack:
                _memo_stack[-1][(t.data_ptr(), tuple(t.shape), t.stride(), k_major, _packed_kind())] = hit
            return hit
    if _memo_stack:
        key = (t.data_ptr(), tuple(t.shape), t.stride(), k_major, _packed_kind())
        hit = _memo_stack[-1].get(key)
        if hit is None:
            hit = _memo_stack[-1][key] = pack3(t, k_major)
        return hit
    return pack3(t, k_major)


def set_gemm_profiling(sink):
    """``sink`` = list to append (flops, start_event, end_event) per lstc_gemm launch, or None to stop."""
    global _gemm_prof
    _gemm_prof = sink


# --------------------------------------------------------------------------------------- helpers
def _mat(t: torch.Tensor):
    """(ptr, rows, cols, ld) of a 2-D f32 tensor whose rows are contiguous."""
    if t.dim() != 2 or t.dtype != torch.float32:
        raise RuntimeError(f"lstc gemm operands are 2-D float32 tensors, got {tuple(t.shape)} {t.dtype}")
    if t.stride(1) != 1 and t.shape[1] != 1:
        raise RuntimeError("lstc gemm operands need unit stride along the last dim")
    return dev_ptr(t), t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


_SMALL_M_SPLIT = os.environ.get("LSTC_SMALL_M_SPLIT", "1") != "0"      # A/B hook: 0 = small products as ONE launch over the whole K range
_SMALL_M_ROWS_AS = None      # test hook: choose the K chunks of a small-row product AS IF it had this many rows (a full batch chunked like one
                             # rank's 256-sequence shard: what tests/test_hip_parity.py compares the summed shard gradients with)
_small_m_depth = 0


class small_m_products:
    """Inside this context ``gemm`` may run a product with few output tiles (the CLS-only last layer and the heads: [sequences, d] x
    [d, d], a rank's 256 sequences = 32 tiles of 128 x 128 on 256 CUs) as K chunks - ONE batched launch into partial results + the
    fixed-order sum and epilogue of ``lstc_splitk_finish`` - instead of one workgroup per tile walking the whole K range (latency
    bound: 123 us for 256 x 2048 x 2048 in fp32, 17 TFLOP/s).  Entered by the TRAINING-mode bodies of those Functions only: how a
    product is chunked depends on its row count, so evaluation keeps the property that a row's scores do not depend on what else
    is in the batch (tests: sharded / pooled evaluation bit-identical to per-video evaluation)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _small_m_depth
        _small_m_depth += 1 if self.on else 0
        return self

    def __exit__(self, *exc):
        global _small_m_depth
        _small_m_depth -= 1 if self.on else 0
        return False


def _small_m_split(M: int, N: int, K: int) -> int:
    """K chunks for a product of few output tiles: the largest power of two <= 16 that keeps the items within one round of the
    512 workgroup slots and every chunk >= 256 deep (whole 64-k steps); 1 = leave it alone (more than a quarter of a round already)."""
    tiles = -(-(_SMALL_M_ROWS_AS or M) // 128) * -(-N // 128)
    if tiles > 128 or K < 512 or N % 4 or M * N >= 1 << 32:       # (2048-row products - 256 tiles - gain nothing in either mode: measured)
        return 1
    s = 1
    while s < 16 and tiles * s * 2 <= 512 and K % (s * 2 * 64) == 0 and K // (s * 2) >= 256:
        s *= 2
    return s


def _gemm_small_m(a, b, trans_a, trans_b, M, N, K, lda, ldb, s, out, bias, relu, dropout, residual, relu_mask, accumulate, alpha):
    """``gemm`` as ``s`` K chunks (one batched launch) + lstc_splitk_finish; same epilogue semantics."""
    dev = a.device
    Kc = K // s
    parts = torch.empty((s, M * N), device=dev, dtype=torch.float32)
    gemm_batched(a, b, parts, M, N, Kc, lda, ldb, N, trans_a, trans_b, s, Kc * lda if trans_a else Kc, Kc if trans_b else Kc * ldb, M * N,
                 alpha=alpha)
    if out is None:
        out = torch.empty((M, N), device=dev, dtype=torch.float32)
    pc, cr, cc, ldc = _mat(out)
    if (cr, cc) != (M, N):
        raise RuntimeError(f"gemm: out is {cr}x{cc}, the product is {M}x{N}")
    flags, p_drop, seed = 0, 0.0, 0
    pbias = pres = pmask = None
    ldr = ldm = 0
    if bias is not None:
        flags |= EPI_BIAS
        pbias = dev_ptr(bias)
    if relu:
        flags |= EPI_RELU
    if dropout is not None and dropout[0] > 0.0:
        flags |= EPI_DROPOUT
        p_drop, seed = float(dropout[0]), int(dropout[1])
    if residual is not None:
        flags |= EPI_RESIDUAL
        pres, rr, rc_, ldr = _mat(residual)
        if (rr, rc_) != (M, N):
            raise RuntimeError(f"gemm: residual is {rr}x{rc_}, the product is {M}x{N}")
    if relu_mask is not None:
        flags |= EPI_RELU_MASK
        pmask, mr, mc, ldm = _mat(relu_mask)
        if (mr, mc) != (M, N):
            raise RuntimeError(f"gemm: relu_mask is {mr}x{mc}, the product is {M}x{N}")
    if accumulate:
        flags |= EPI_ACCUM
    check(_lib.load().lstc_splitk_finish(dev_ptr(parts), s, M * N, M, N, pbias, pres, ldr, pmask, ldm, pc, ldc, flags, p_drop, seed,
                                         1, 0, 0, stream_ptr()), "lstc_splitk_finish")
    return out


def _head_k_chunks(w: torch.Tensor, H: int, dh: int, dm: int, s: int) -> torch.Tensor:
    """``w`` [H * dh, dm] as [H, s, dh, dm / s] (every head's K chunks back to back, so chunk z = h * s + kc of a batched launch is
    at z * dh * (dm / s)); kept until the weight changes (never across a graph capture: the copy must be part of the graph)."""
    key = (s, _wstamp(w), w._version, w.data_ptr())       # raw-pointer updates bump the stamp, torch's in-place ops the version
    c = w.__dict__.get("_lstc_kchunks")
    if c is None or c[0] != key or torch.cuda.is_current_stream_capturing():
        c = (key, w.detach().contiguous().view(H, dh, s, dm // s).permute(0, 2, 1, 3).contiguous())
        w.__dict__["_lstc_kchunks"] = c
    return c[1]


def per_head_rows_wT(a3: torch.Tensor, w: torch.Tensor, out: torch.Tensor, N: int, dh: int, dm: int, H: int, alpha: float = 1.0):
    """out[n, h * dh + j] = alpha * sum_c a3[n, h, c] * w[h * dh + j, c]: the per-head products x-bar_h W_v,h^T and du_h W_k,h^T of the
    re-associated CLS attention (one batched launch over the heads).  With few sequences (small_m_products) the 4 H tiles of a
    rank's 256 sequences walk K = d_model alone: then every head's K range is chunked as well - batch z = h * s + kc over A as
    it lies, a [H, s, dh, K/s] arrangement of the weight and partials [H, s, N, dh] - and lstc_splitk_finish sums the groups."""
    s = 1
    if _small_m_depth > 0 and _SMALL_M_SPLIT and dh % 4 == 0:
        tiles = H * -(-(_SMALL_M_ROWS_AS or N) // 128) * -(-dh // 128)
        if tiles <= 128 and dm >= 512:
            while s < 16 and tiles * s * 2 <= 512 and dm % (s * 2 * 64) == 0 and dm // (s * 2) >= 256:
                s *= 2
    if s == 1:
        return gemm_batched(a3, w, out, N, dh, dm, H * dm, dm, H * dh, False, True, H, dm, dh * dm, dh, alpha=alpha)
    Kc = dm // s
    wc = _head_k_chunks(w, H, dh, dm, s)
    parts = torch.empty((H * s, N * dh), device=a3.device, dtype=torch.float32)
    gemm_batched(a3, wc, parts, N, dh, Kc, H * dm, Kc, dh, False, True, H * s, Kc, dh * Kc, N * dh, alpha=alpha)
    check(_lib.load().lstc_splitk_finish(dev_ptr(parts), s, N * dh, N, dh, None, None, 0, None, 0, dev_ptr(out), H * dh, 0, 0.0, 0,
                                         H, s * N * dh, dh, stream_ptr()), "lstc_splitk_finish")
    return out


def _small_m_fwd(fn):
    """forward(ctx, x, ..., cfg) of a Function whose products may have few rows - the CLS-only last layer (x = [sequences, d] for its
    FFN and the heads; the attention Functions of that layer by construction): small_m_products while the module is in training
    mode.  A full layer's [N, S, d] input never enters the context (its products are large, and at reduced test sizes the fused /
    unfused pack paths must keep routing alike)."""
    @functools.wraps(fn)
    def wrapper(ctx, *args):
        on = bool(args[-1].get("training", False)) and (fn.__qualname__.startswith(("MHACls", "HeadFunction")) or args[0].dim() == 2)
        ctx._small_m = on
        with small_m_products(on):
            return fn(ctx, *args)
    return wrapper


def _small_m_bwd(fn):
    @functools.wraps(fn)
    def wrapper(ctx, *args):
        with small_m_products(bool(getattr(ctx, "_small_m", False))):
            return fn(ctx, *args)
    return wrapper


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a=False, trans_b=False, out: Optional[torch.Tensor] = None,
         bias=None, relu=False, dropout=None, residual=None, relu_mask=None, accumulate=False, alpha=1.0,
         split_k=1, variant=0, out_pack=False) -> torch.Tensor:
    """``out = epi(alpha * op(a) @ op(b))`` through ``lstc_gemm`` (include/lstc_hip.h).  ``a`` / ``b`` may be ``Packed``
    operands (f32x3 mode): then they stand for the logical [M, K] / [N, K] matrices and trans_a / trans_b are moot."""
    dev = (a.buf if isinstance(a, Packed) else a).device
    if isinstance(a, Packed):
        M, K, lda, pa = a.rows, a.K, a.K, None
    else:
        pa, ar, ac, lda = _mat(a)
        M, K = (ac, ar) if trans_a else (ar, ac)
    if isinstance(b, Packed):
        N, Kb, ldb, pb = b.rows, b.K, b.K, None
    else:
        pb, br, bc, ldb = _mat(b)
        Kb, N = (bc, br) if trans_b else (br, bc)
    if K != Kb:
        raise RuntimeError(f"gemm inner dims differ: {K} vs {Kb}")
    # (not the products that feed a ReLU: which side of zero a pre-activation of magnitude 1e-7 lands on follows the summation order,
    # and one flipped unit of the CLS-only layer's FFN rewrites a row of dW1 - they keep the k order of the one-launch form, which is
    # also the evaluation's)
    if _small_m_depth > 0 and _SMALL_M_SPLIT and pa is not None and pb is not None and split_k == 1 and variant == 0 and not out_pack \
            and not relu and not isinstance(residual, Packed) and not isinstance(relu_mask, Packed) and lda % 4 == 0 and ldb % 4 == 0:
        s_small = _small_m_split(M, N, K)
        if s_small > 1:
            return _gemm_small_m(a, b, trans_a, trans_b, M, N, K, lda, ldb, s_small, out, bias, relu, dropout, residual, relu_mask,
                                 accumulate, alpha)
    dtype = _compute_dtype
    pkind = _packed_kind()
    packed = False
    if pkind is not None:
        # small products (heads of 512 / 32 columns, the CLS-only last layer) stay on the exact-f32 kernel (f32x3 mode) or on
        # the convert-while-staging bf16 kernel (bf16 mode)
        # (bf16 mode: a quarter of the size suffices since round 6 - products of a few 256 x 256 tiles run as 128 x 128 quarter items
        # on the packed kernel (csrc/gemm_bf16p.hip, gemm_bf16p_q_kernel): the head's 256 x 512 x 2048 product of one rank of the 8-GPU
        # split 0.084 ms on the convert-while-staging kernel, four workgroups walking K alone)
        min_mnk = min(_x3_min[2], _BF16P_MIN_MNK) if pkind == _lib.BF16P else _x3_min[2]
        if isinstance(a, Packed) or isinstance(b, Packed) or (min(M, N) >= _x3_min[0] and K >= _x3_min[1] and M * N * K >= min_mnk):
            a = _packed_operand(a, trans_a)
            b = _packed_operand(b, not trans_b)
            pa, pb = dev_ptr(a.buf), dev_ptr(b.buf)
            dtype, packed = pkind, True
        elif dtype == _lib.F32X3:
            dtype = F32
    parts = None
    if out_pack:
        # bf16 mode: the result is written ONLY as the packed bf16 operand of the products that consume it (LSTC_EPI_OUT_PACK)
        if not (packed and dtype == _lib.BF16P and packed_out_shape(M, N)) or out is not None or split_k > 1 or accumulate:
            raise RuntimeError(f"gemm(out_pack=True): [{M}, {N}] x K={K} does not qualify (bf16 packed product on whole 256-tiles)")
        pbuf = torch.empty((int(_lib.load().lstc_pack1_bytes(M, N)),), device=dev, dtype=torch.uint8)
    if out is None and not out_pack:
        if split_k > 1 and packed and _DETERMINISTIC_WGRAD:
            # K splits of the packed kernel into separate partials, summed in a fixed order afterwards (no atomics).  The
            # library launches lstc_gemm_splits() slices, possibly fewer than asked for: size and sum exactly that many
            split_k = int(_lib.load().lstc_gemm_splits(dtype, K, split_k))
            parts = torch.empty((split_k, M * N), device=dev, dtype=torch.float32)
            out = parts[0].view(M, N)
        else:
            out = torch.empty((M, N), device=dev, dtype=torch.float32)
            if split_k > 1:
                out.zero_()
    if out_pack:
        pc, ldc = dev_ptr(pbuf), N
    else:
        pc, cr, cc, ldc = _mat(out)
        if (cr, cc) != (M, N):
            raise RuntimeError(f"gemm: out is {cr}x{cc}, the product is {M}x{N}")
    d = GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, lda, ldb, ldc
    d.transA, d.transB, d.dtype = int(trans_a), int(trans_b), dtype
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
        d.bias = dev_ptr(bias)
    if relu:
        flags |= EPI_RELU
    if dropout is not None and dropout[0] > 0.0:
        flags |= EPI_DROPOUT
        d.dropout_p, d.dropout_seed = float(dropout[0]), int(dropout[1])
    if isinstance(residual, Packed):
        if (residual.rows, residual.K, residual.kind) != (M, N, _lib.BF16P) or not out_pack:
            raise RuntimeError(f"gemm: packed residual is {residual.rows}x{residual.K} (kind {residual.kind}), the product is {M}x{N}"
                               f" (out_pack={out_pack})")
        flags |= EPI_RESIDUAL | EPI_RESIDUAL_PACK
        d.residual, d.ldr = dev_ptr(residual.buf), N
    elif residual is not None:
        flags |= EPI_RESIDUAL
        pr, rr, rc_, ldr = _mat(residual)
        if (rr, rc_) != (M, N):
            raise RuntimeError(f"gemm: residual is {rr}x{rc_}, the product is {M}x{N}")
        d.residual, d.ldr = pr, ldr
    if isinstance(relu_mask, Packed):
        if (relu_mask.rows, relu_mask.K, relu_mask.kind) != (M, N, _lib.BF16P):
            raise RuntimeError(f"gemm: packed relu_mask is {relu_mask.rows}x{relu_mask.K} (kind {relu_mask.kind}), the product is {M}x{N}")
        flags |= EPI_RELU_MASK | EPI_RELU_MASK_PACK
        d.relu_src, d.ld_relu = dev_ptr(relu_mask.buf), N
    elif relu_mask is not None:
        flags |= EPI_RELU_MASK
        pm, mr, mc, ldm = _mat(relu_mask)
        if (mr, mc) != (M, N):
            raise RuntimeError(f"gemm: relu_mask is {mr}x{mc}, the product is {M}x{N}")
        d.relu_src, d.ld_relu = pm, ldm
    if accumulate:
        flags |= EPI_ACCUM
    if out_pack:
        flags |= EPI_OUT_PACK
    d.flags, d.alpha, d.split_k, d.variant = flags, float(alpha), int(split_k), int(variant)
    d.A, d.B, d.C = pa, pb, pc
    if packed:
        d.transA, d.transB = 0, 1              # packs of [M, K] and [N, K]
        if parts is not None:
            d.batch_stride_c = M * N
    _launch_gemm(d, 2.0 * M * N * K)
    if out_pack:
        return Packed(pbuf, M, N, _lib.BF16P)
    if parts is not None:
        return colsum(parts).view(M, N)
    return out


def packed_out_shape(M: int, N: int) -> bool:
    """[M, N] results that the bf16 packed kernel can emit as a packed operand (include/lstc_hip.h, LSTC_EPI_OUT_PACK)."""
    if _small_m_depth > 0 and _SMALL_M_SPLIT and -(-M // 128) * -(-N // 128) <= 128:
        return False            # a small product of the CLS-only layer / the heads: its consumers run as K chunks on f32 operands
    return (_FUSE_PACKS and _packed_kind() == _lib.BF16P and M % 256 == 0 and N % 256 == 0 and
            M >= max(_x3_min[0], 1) and N >= max(_x3_min[1], 256) and M * N * max(_x3_min[0], 256) >= _x3_min[2])


def colsum_pack(pk: Packed) -> torch.Tensor:
    """Column sums [K] of a packed bf16 operand [rows, K] (lstc_colsum_pack1)."""
    n_partial = int(min((pk.rows + 127) // 128, 64))
    partial = torch.empty((n_partial + 2, pk.K), device=pk.buf.device, dtype=torch.float32)
    out = torch.empty((pk.K,), device=pk.buf.device, dtype=torch.float32)
    check(_lib.load().lstc_colsum_pack1(dev_ptr(pk.buf), pk.rows, pk.K, dev_ptr(partial), n_partial, dev_ptr(out), 0, stream_ptr()),
          "lstc_colsum_pack1")
    return out


def _launch_gemm(d, flops):
    if _gemm_prof is not None:                 # bench.py: HIP events on the launch stream around every GEMM
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().lstc_gemm(C.byref(d), stream_ptr()), "lstc_gemm")
        e1.record()
        _gemm_prof.append((flops, e0, e1))
        return
    check(_lib.load().lstc_gemm(C.byref(d), stream_ptr()), "lstc_gemm")


def maybe_pack(t: torch.Tensor):
    """Packed form of an activation [rows, K] when f32x3 mode would route its products to the packed kernel, else None.
    Lets a forward body pack X once, feed several GEMMs and keep the pack for the weight gradient of the backward."""
    if _packed_kind() is None or t.shape[0] < max(_x3_min[0], 1) or t.shape[1] < _x3_min[1]:
        return None
    if t.shape[0] * t.shape[1] * max(_x3_min[0], 1) < _x3_min[2]:
        return None
    return _packed_operand(t, False)


def _wgrad_split(m_out: int, n_out: int, tile: int = 128) -> int:
    """Split-K factor for weight gradients: K = token count is huge and the output small, so the K range is split until the
    grid fills the chip (256 CUs; 128x128 tiles at two workgroups per CU, 256x256 for the persistent packed bf16 kernel at one)
    - and so that the items fill WHOLE rounds of those slots: the fused dW_qkv of bf16 mode is 24 x 8 = 192 tiles, i.e. 1.5
    rounds of 256 with two splits (the second round half empty: 978 TFLOP/s) but exactly 3 rounds with four (measured 5.16 ->
    4.2 ms per launch).  Smallest power of two <= 16 whose last round is at least 90 % full, else the best-filled one."""
    tiles = ((m_out + tile - 1) // tile) * ((n_out + tile - 1) // tile)
    slots = 256 if tile == 256 else 512     # the 256x256 kernel holds one workgroup per CU (128 KB of LDS), the others two
    best, best_eff = 1, 0.0
    for s in (1, 2, 4, 8, 16):              # powers of two: the token count of every config divides by them (batched f32 partials)
        items = tiles * s
        if items * 2 < slots and s < 16:
            continue                        # not even half a round yet
        eff = (items / slots) / -(-items // slots)
        if eff >= 0.9:
            return s
        if eff > best_eff + 1e-9:
            best, best_eff = s, eff
    return best


def _wgrad_split_bf16p(m_out: int, n_out: int, tokens: int) -> int:
    """Split-K factor of a weight gradient on the packed bf16 kernel (256 x 256 tiles, one workgroup per CU) from a cost model instead
    of round filling alone: rounds(s) x (K steps per item x 1.84 us + ~8 us of item prologue / partial-tile epilogue) + the ordered sum
    of the s partials ((s + 1) x 4 m n bytes at ~4 TB/s; none for s = 1).  At the headline token counts it picks what `_wgrad_split`
    picks (the K loops dominate: 4 / 2 / 4 / 8 for dW_qkv, dW_1 / dW_2, dW_fc and UBnormal's dW_qkv); with a few thousand tokens - one
    rank of an 8-GPU split of the STN / UCF configs - the fixed costs and the partial sums decide: the STN rank's [3072, 2048] x 4352
    tokens ran 8 splits of 8.5 K steps each at 576 TFLOP/s (profiles/r06_rank_gemm_launch_table_bf16.txt era), two splits model 1.5x faster."""
    tiles = -(-m_out // 256) * -(-n_out // 256)
    steps = -(-tokens // 64)
    best, best_t = 1, None
    for s in (1, 2, 4, 8, 16):
        if s > 1 and steps // s < 8:
            break
        rounds = -(-(tiles * s) // 256)
        t = rounds * (-(-steps // s) * 1.84 + 8.0)
        if s > 1:
            t += (s + 1) * 4.0 * m_out * n_out / 4.0e6
        if best_t is None or t < best_t - 1e-9:
            best, best_t = s, t
    return best


# ---- gradient sinks (data parallel).  dist.GradAllReducer keeps every gradient of a bucket in ONE flat buffer that RCCL reduces in
# place.  Letting autograd ACCUMULATE into views of that buffer costs a 407-MB fill per step plus a read-add-write of every weight
# gradient after the kernel that produced it.  Instead the reducer hangs a sink on each large weight -
# ``w._lstc_grad_sink = (view of the bucket shaped like w, done(w))`` - and the Function bodies below write the weight
# gradient THERE (``wgrad(..., out=grad_sink(w))``: the GEMM's C, or the fixed-order sum of its split-K partials), tell the reducer
# (``deliver`` -> ``done``: the bucket's all-reduce starts when its last gradient has been issued) and return None to autograd.
# Same kernels, same arithmetic, same values as the plain path (0 + g = g): only the destination differs.
def grad_sink(w):
    """Destination view for the gradient of weight ``w``, or None (no reducer: the gradient goes back through autograd)."""
    s = _live_sink(w)
    return None if s is None else s[0]


def _live_sink(w):
    """The sink of ``w`` while it is in force: the reducer keeps ``w.grad`` pointing at the sink's view; once somebody else
    owns ``w.grad`` (``optimizer.zero_grad(set_to_none=True)``, a model reused without its reducer) the sink is ignored and the
    gradient travels through autograd again."""
    s = w.__dict__.get("_lstc_grad_sink") if w is not None else None
    if s is None or w.grad is None or w.grad.data_ptr() != s[0].data_ptr():
        return None
    return s


def fused_grad_sink(wq, wk, wv):
    """One [rows_q + rows_k + rows_v, d] destination when the three sinks are consecutive in their bucket (they are: the
    reducer lays weights out in ``parameters()`` order), else None."""
    sq, sk, sv = grad_sink(wq), grad_sink(wk), grad_sink(wv)
    if sq is None or sk is None or sv is None or not (sq.is_contiguous() and sk.is_contiguous() and sv.is_contiguous()):
        return None
    if sk.data_ptr() != sq.data_ptr() + 4 * sq.numel() or sv.data_ptr() != sk.data_ptr() + 4 * sk.numel() or \
            not (sq.shape[1] == sk.shape[1] == sv.shape[1]):
        return None
    return torch.as_strided(sq, (sq.shape[0] + sk.shape[0] + sv.shape[0], sq.shape[1]), (sq.shape[1], 1), sq.storage_offset())


def deliver(w, g):
    """What a Function.backward returns for the gradient ``g`` of weight ``w``: ``g`` itself, or - when ``w`` carries a sink -
    None after making sure the sink holds ``g`` (a no-op when the producing kernel already wrote there) and notifying the reducer."""
    s = _live_sink(w)
    if s is None or g is None:
        return g
    view, done = s
    if g.data_ptr() != view.data_ptr() or tuple(g.shape) != tuple(view.shape) or not g.is_contiguous():
        view.copy_(g)                         # e.g. the real rows of a padded-hidden gradient: one strided copy, no fill, no add
    done(w)
    return None


def wgrad(dy: torch.Tensor, x: torch.Tensor, x_pack: Optional[Packed] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[out, in] = dy[T, out]^T @ x[T, in]  (autograd of nn.Linear's weight).  f32x3 mode: the contraction runs over the
    tokens, i.e. along the ROWS of the packs that the forward (X) and input-gradient (dY) products already made, so those
    packs are reused through the transposed-read form of the packed kernel (``x_pack`` = the forward's pack of ``x``).
    ``out``: optional contiguous [out, in] f32 destination (a gradient sink); the atomic split-K forms ignore it."""
    T, O = (dy.rows, dy.K) if isinstance(dy, Packed) else dy.shape
    I = x.shape[1] if x is not None else x_pack.K
    if out is not None and (tuple(out.shape) != (O, I) or not out.is_contiguous() or out.dtype != torch.float32):
        raise RuntimeError(f"wgrad: out must be a contiguous float32 [{O}, {I}] tensor")
    pkind = _packed_kind()
    # split-K factor: 256x256 output tiles on the packed bf16 kernel, 128x128 everywhere else
    if pkind == _lib.BF16P and _WGRAD_COST_MODEL:
        s = _wgrad_split_bf16p(O, I, T) if T >= 4096 else 1       # (below 4096 tokens: one launch, no partial buffers - as rounds 2 - 5)
    else:
        s = _wgrad_split(O, I, 256 if pkind == _lib.BF16P else 128) if T >= 4096 else 1
    tr_ok = T % 128 == 0 and (pkind == _lib.BF16P or (O % 128 == 0 and I % 128 == 0))
    # a gradient that already IS a packed operand (layernorm_bwd_branch emits df packed whenever [rows, d_model] fills the tile
    # grid, whatever the width of its partner) always takes the packed TR form; the partner is packed on demand (the TR kernel
    # handles ragged O / I) - e.g. n_hidden < 256 or H*d_v < 256, where the forward kept no pack of the hidden / of O
    if pkind is not None and tr_ok and \
            (x_pack is not None or (isinstance(dy, Packed) and x is not None and dy.kind == pkind) or
             (min(O, I) >= _x3_min[0] and T >= _x3_min[1] and T * O * I >= _x3_min[2])):
        ap = _packed_operand(dy, False)
        bp = x_pack if x_pack is not None else _packed_operand(x, False)
        dev = ap.buf.device
        # K splits write separate partials that lstc_colsum adds in a fixed order (no atomics: bit-reproducible).  The
        # library may launch fewer slices than asked for (132 K tiles / 16 -> 15 slices): allocate and sum exactly those,
        # an extra row would add uninitialised memory into the gradient
        s = int(_lib.load().lstc_gemm_splits(pkind, T, s))
        det = s > 1 and _DETERMINISTIC_WGRAD
        if s == 1 and out is not None:
            part = out.view(1, O * I)
        else:
            part = torch.empty((s, O * I), device=dev, dtype=torch.float32) if (det or s == 1) else \
                torch.zeros((1, O * I), device=dev, dtype=torch.float32)
        d = GemmDesc()
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc = O, I, T, O, I, I
        d.transA, d.transB, d.dtype, d.flags, d.alpha, d.split_k = 1, 0, pkind, 0, 1.0, s
        d.batch_stride_c = O * I if (s > 1 and _DETERMINISTIC_WGRAD) else 0
        d.A, d.B, d.C = dev_ptr(ap.buf), dev_ptr(bp.buf), dev_ptr(part)
        _launch_gemm(d, 2.0 * O * I * T)
        return (colsum(part, out=None if out is None else out.view(O * I)) if det else part).view(O, I)
    if isinstance(dy, Packed):
        raise RuntimeError(f"wgrad: packed gradient [{T}, {O}] x [{T}, {I}] does not qualify for the packed kernel")
    x3_big = pkind is not None and min(O, I) >= _x3_min[0] and T >= _x3_min[1] and T * O * I >= _x3_min[2]
    if pkind == _lib.BF16P and not x3_big:
        s = _wgrad_split(O, I, 128) if T >= 4096 else 1       # not a packed product after all: 128x128-tile kernel
    if s > 1 and (_compute_dtype == F32 or (_compute_dtype == _lib.F32X3 and not x3_big)) and _DETERMINISTIC_WGRAD and \
            T % s == 0 and dy.stride(1) == 1 and x.stride(1) == 1:
        # exact-f32 kernel (also the small products of f32x3 mode): the s K-chunks are ONE batched launch into [s, O, I] partials, summed in a fixed order by
        # lstc_colsum - same parallelism as the atomic split-K, but the step is bit-reproducible run to run
        part = torch.empty((s, O * I), device=dy.device, dtype=torch.float32)
        Tc = T // s
        gemm_batched(dy, x, part, O, I, Tc, dy.stride(0), x.stride(0), I, True, False, s, Tc * dy.stride(0), Tc * x.stride(0), O * I)
        return colsum(part, out=None if out is None else out.view(O * I)).view(O, I)
    return gemm(dy, x, trans_a=True, trans_b=False, split_k=s, out=out if s == 1 else None)


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate=False) -> torch.Tensor:
    px, rows, cols, ld = _mat(x)
    n_partial = int(min(rows, 128))
    partial = torch.empty((n_partial, cols), device=x.device, dtype=torch.float32)
    if out is None:
        out = torch.empty((cols,), device=x.device, dtype=torch.float32)
    check(_lib.load().lstc_colsum(px, rows, cols, ld, dev_ptr(partial), n_partial, dev_ptr(out), int(accumulate),
                                  stream_ptr()), "lstc_colsum")
    return out


def colsum_planes(x3: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
    """Column sums of the first ``n`` planes of a contiguous [planes, rows, cols] tensor -> [n, cols], in TWO launches whatever
    ``n`` (lstc_colsum_batched); row ``b`` is bit-identical to ``colsum(x3[b])`` when that takes the two-pass form."""
    planes, rows, cols = x3.shape
    n = planes if n is None else n
    out = torch.empty((n, cols), device=x3.device, dtype=torch.float32)
    if rows <= 12 or not x3.is_contiguous():            # colsum's one-pass small-row form: keep its arithmetic
        for b in range(n):
            colsum(x3[b], out=out[b])
        return out
    npart = int(min(rows, 128))
    partial = torch.empty((n, npart, cols), device=x3.device, dtype=torch.float32)
    check(_lib.load().lstc_colsum_batched(dev_ptr(x3), n, rows, cols, cols, rows * cols, dev_ptr(partial), npart, dev_ptr(out),
                                          stream_ptr()), "lstc_colsum_batched")
    return out


def dropout_apply(x: torch.Tensor, p: float, seed: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().lstc_dropout_apply(dev_ptr(x), dev_ptr(out), x.numel(), float(p), int(seed), stream_ptr()),
          "lstc_dropout_apply")
    return out


def dropout_apply_pack(x: "Packed", p: float, seed: int) -> "Packed":
    """Dropout replay on a packed bf16 operand (lstc_dropout_apply_pack): the mask of the flat index row * K + col."""
    buf = torch.empty_like(x.buf)
    check(_lib.load().lstc_dropout_apply_pack(dev_ptr(x.buf), dev_ptr(buf), x.rows, x.K, float(p), int(seed), stream_ptr()),
          "lstc_dropout_apply_pack")
    return Packed(buf, x.rows, x.K, x.kind)


def gather_rows(bank: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = bank[idx[r]] over the leading dimension (``lstc_gather_rows``): forms a training batch from an
    HBM-resident feature bank.  ``idx`` int64 on the same device; trailing dims of ``bank`` must hold a multiple of 4 floats."""
    if bank.dtype != torch.float32 or idx.dtype != torch.int64 or not bank.is_contiguous():
        raise TypeError("gather_rows: bank must be contiguous float32 and idx int64")
    idx = idx.contiguous()
    row = int(bank[0].numel())
    if out is None:
        out = torch.empty((idx.numel(),) + tuple(bank.shape[1:]), device=bank.device, dtype=torch.float32)
    check(_lib.load().lstc_gather_rows(dev_ptr(bank), bank.shape[0], dev_ptr(idx), dev_ptr(out), idx.numel(), row,
                                       stream_ptr()), "lstc_gather_rows")
    return out


def dropout_mask(shape, p: float, seed: int, device) -> torch.Tensor:
    m = torch.empty(shape, device=device, dtype=torch.uint8)
    check(_lib.load().lstc_dropout_mask(dev_ptr(m), m.numel(), float(p), int(seed), stream_ptr()), "lstc_dropout_mask")
    return m


def layernorm_fwd(x2: torch.Tensor, gamma, beta, eps=1e-6, pack=False):
    """``pack``: the result is the input of another encoder block - in bf16 mode the kernel also writes its packed bf16 form,
    which the block's first GEMM picks up (``_producer_pack``) instead of packing the f32 rows again."""
    rows, d = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty((rows,), device=x2.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    lib = _lib.load()
    if pack and _fused_pack_shape(rows, d):
        buf = torch.empty((int(lib.lstc_pack1_bytes(rows, d)),), device=x2.device, dtype=torch.uint8)
        check(lib.lstc_layernorm_fwd_pack(dev_ptr(x2), dev_ptr(gamma), dev_ptr(beta), dev_ptr(y), dev_ptr(mean), dev_ptr(rstd),
                                          rows, d, float(eps), dev_ptr(buf), stream_ptr()), "lstc_layernorm_fwd_pack")
        _register_pack(y, Packed(buf, rows, d, _lib.BF16P))
        return y, mean, rstd
    check(lib.lstc_layernorm_fwd(dev_ptr(x2), dev_ptr(gamma), dev_ptr(beta), dev_ptr(y), dev_ptr(mean),
                                 dev_ptr(rstd), rows, d, float(eps), stream_ptr()), "lstc_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy2, x2, gamma, mean, rstd):
    rows, d = x2.shape
    dy2 = dy2.contiguous()
    dx = torch.empty_like(x2)
    # one partial row per workgroup: 4 resident workgroups per CU at the model widths (two waves per row, csrc/rowops.hip), 2 else
    n_partial = int(min(max(rows // 4, 1), 1024 if d in (512, 1024, 2048) else 512))
    partial = torch.empty((2, n_partial, d), device=x2.device, dtype=torch.float32)
    check(_lib.load().lstc_layernorm_bwd(dev_ptr(dy2), dev_ptr(x2), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd),
                                         dev_ptr(dx), dev_ptr(partial), n_partial, rows, d, stream_ptr()),
          "lstc_layernorm_bwd")
    sums = colsum_planes(partial)
    return dx, sums[0], sums[1]


def layernorm_bwd_branch(dz2, y, gamma, mean, rstd, p: float, seed: int, layer_norm: bool, want_bias: bool):
    """Backward of ``z = [LayerNorm](dropout(f) + x)`` up to the two things its callers consume: ``dy`` (gradient of the
    residual sum, f32: it flows on into x) and ``df`` (gradient of the dropout's input f: operand of the weight / input
    gradient GEMMs of the Linear that produced f).  Returns (dy, df, dgamma, dbeta, dbias) with dbias = column sums of df when
    ``want_bias``.  bf16 mode with LayerNorm on a tile-filling shape: ONE kernel writes dy, the packed bf16 df (dropout
    replayed in registers) and the partial sums - ``df`` is then a ``Packed``; otherwise lstc_layernorm_bwd +
    lstc_dropout_apply (+ lstc_colsum)."""
    rows, d = dz2.shape
    if layer_norm and _fused_pack_shape(rows, d) and rows * d <= 0xffffffff:
        lib = _lib.load()
        dz2 = dz2.contiguous()
        dy = torch.empty_like(y)
        n_partial = int(min(max(rows // 4, 1), 768))        # 3 workgroups per CU resident (csrc/rowops.hip, ln_bwd_pack2)
        partial = torch.empty((3, n_partial, d), device=y.device, dtype=torch.float32)
        buf = torch.empty((int(lib.lstc_pack1_bytes(rows, d)),), device=y.device, dtype=torch.uint8)
        check(lib.lstc_layernorm_bwd_drop_pack(dev_ptr(dz2), dev_ptr(y), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd),
                                               dev_ptr(dy), dev_ptr(partial), n_partial, rows, d, float(p), int(seed),
                                               dev_ptr(buf), stream_ptr()), "lstc_layernorm_bwd_drop_pack")
        sums = colsum_planes(partial, 3 if want_bias else 2)        # dgamma, dbeta(, dbias): one batched two-pass sum
        return dy, Packed(buf, rows, d, _lib.BF16P), sums[0], sums[1], (sums[2] if want_bias else None)
    if layer_norm and p > 0 and _FUSE_PACKS and d in (512, 1024, 2048) and rows * d <= 0xffffffff:
        # f32 modes: the same kernel writes the dropped gradient as a second f32 result (no lstc_dropout_apply pass, and the
        # bias gradient's column sums come out of the same pass)
        lib = _lib.load()
        dz2 = dz2.contiguous()
        dy, df = torch.empty_like(y), torch.empty_like(y)
        n_partial = int(min(max(rows // 4, 1), 768))
        partial = torch.empty((3, n_partial, d), device=y.device, dtype=torch.float32)
        check(lib.lstc_layernorm_bwd_drop(dev_ptr(dz2), dev_ptr(y), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd), dev_ptr(dy),
                                          dev_ptr(df), dev_ptr(partial), n_partial, rows, d, float(p), int(seed), stream_ptr()),
              "lstc_layernorm_bwd_drop")
        sums = colsum_planes(partial, 3 if want_bias else 2)
        return dy, df, sums[0], sums[1], (sums[2] if want_bias else None)
    dgamma = dbeta = None
    if layer_norm:
        dy, dgamma, dbeta = layernorm_bwd(dz2, y, gamma, mean, rstd)
    else:
        dy = dz2
    df = dropout_apply(dy, p, seed) if p > 0 else dy
    return dy, df, dgamma, dbeta, (colsum(df) if want_bias else None)


def layernorm_fwd_act(x, gamma, beta, eps=1e-6, want_f32=False, want_pack=True):
    """LayerNorm of the bf16 stream (lstc_layernorm_fwd_act): ``x`` a ``Packed`` [rows, d] (or an f32 [rows, d] tensor); returns
    (y f32 | None, y Packed | None, mean, rstd)."""
    if isinstance(x, Packed):
        rows, d, dev, px, pxp = x.rows, x.K, x.buf.device, None, dev_ptr(x.buf)
    else:
        rows, d = x.shape
        dev, px, pxp = x.device, dev_ptr(x.contiguous()), None
    lib = _lib.load()
    y = torch.empty((rows, d), device=dev, dtype=torch.float32) if want_f32 else None
    ybuf = torch.empty((int(lib.lstc_pack1_bytes(rows, d)),), device=dev, dtype=torch.uint8) if want_pack else None
    mean = torch.empty((rows,), device=dev, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    check(lib.lstc_layernorm_fwd_act(px, pxp, dev_ptr(gamma), dev_ptr(beta), dev_ptr(y), dev_ptr(ybuf), dev_ptr(mean), dev_ptr(rstd),
                                     rows, d, float(eps), stream_ptr()), "lstc_layernorm_fwd_act")
    return y, (Packed(ybuf, rows, d, _lib.BF16P) if want_pack else None), mean, rstd


def layernorm_bwd_act(dz, y: Packed, gamma, mean, rstd, p: float, seed: int, want_dx: bool, want_bias: bool):
    """Backward of ``z = LayerNorm(dropout(f) + x)`` on the bf16 stream (lstc_layernorm_bwd_act).  ``dz``: ``Packed`` or f32
    [rows, d]; ``y``: the pack of the pre-LayerNorm sum.  Returns (dy Packed | None, df Packed, dgamma, dbeta, dbias | None)."""
    rows, d, dev = y.rows, y.K, y.buf.device
    lib = _lib.load()
    if isinstance(dz, Packed):
        if (dz.rows, dz.K) != (rows, d):
            raise RuntimeError(f"layernorm_bwd_act: gradient pack is [{dz.rows}, {dz.K}], the activation [{rows}, {d}]")
        pdz, pdzp = None, dev_ptr(dz.buf)
    else:
        dz = dz.contiguous()
        if tuple(dz.shape) != (rows, d) or dz.dtype != torch.float32:
            raise RuntimeError(f"layernorm_bwd_act: gradient is {tuple(dz.shape)} {dz.dtype}, the activation [{rows}, {d}]")
        pdz, pdzp = dev_ptr(dz), None
    nbytes = int(lib.lstc_pack1_bytes(rows, d))
    dxbuf = torch.empty((nbytes,), device=dev, dtype=torch.uint8) if want_dx else None
    dfbuf = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
    n_partial = int(min(max(rows // 4, 1), 768))
    partial = torch.empty((3, n_partial, d), device=dev, dtype=torch.float32)
    check(lib.lstc_layernorm_bwd_act(pdz, pdzp, dev_ptr(y.buf), dev_ptr(gamma), dev_ptr(mean), dev_ptr(rstd), dev_ptr(dxbuf),
                                     dev_ptr(partial), n_partial, rows, d, float(p), int(seed), dev_ptr(dfbuf), stream_ptr()),
          "lstc_layernorm_bwd_act")
    sums = colsum_planes(partial, 3 if want_bias else 2)
    return ((Packed(dxbuf, rows, d, _lib.BF16P) if want_dx else None), Packed(dfbuf, rows, d, _lib.BF16P), sums[0], sums[1],
            (sums[2] if want_bias else None))


def _attn_dtype():
    """LstcAttnDesc.dtype of the current compute mode: bf16 mode contracts bf16-rounded operands on the bf16 MFMA."""
    return _lib.BF16 if (_compute_dtype == _lib.BF16 and not _ATTN_F32) else F32


def attn_fwd_pack(N, S, H, dv) -> bool:
    """bf16 mode: the attention forward can write O as a packed bf16 operand (include/lstc_hip.h, O_pack)."""
    M = N * S
    return (_FUSE_PACKS and _packed_kind() == _lib.BF16P and dv % 32 == 0 and M % 256 == 0 and (H * dv) % 64 == 0 and
            M * H * dv * 2 < 2 ** 31 and M >= max(_x3_min[0], 1) and H * dv >= max(_x3_min[1], 256) and
            M * H * dv * max(_x3_min[0], 256) >= _x3_min[2])


def _qkv_pack_desc(d, qkv: "Packed", H, dk, dv):
    """Packed-input form of LstcAttnDesc: Q | K | V are the column blocks of ONE pack (the fused projection's output)."""
    if not (qkv.kind == _lib.BF16P and qkv.K == H * (2 * dk + dv)):
        raise RuntimeError(f"packed Q | K | V operand: expected a bf16 pack of {H * (2 * dk + dv)} columns, got kind {qkv.kind}, {qkv.K} columns")
    d.in_pack_cols, d.Q_col0, d.K_col0, d.V_col0 = qkv.K, 0, H * dk, 2 * H * dk
    d.Q = d.K = d.V = dev_ptr(qkv.buf)


def attn_packed_inputs(N, S, H, dk, dv) -> bool:
    """bf16 mode: the attention core can read Q | K | V (and dO) as packed bf16 operands (include/lstc_hip.h, in_pack_cols)."""
    return (attn_fwd_pack(N, S, H, dv) and attn_bwd_packs(N, S, H, dk, dv) and _ATTN_PACKED_IN and not _ATTN_F32 and S <= 96 and
            dk % 64 == 0 and dv % 64 == 0 and N * S * H * (2 * dk + dv) * 2 < 2 ** 31)


def attn_fwd(q, k, v, N, S, H, dk, dv, table, index, p_drop, seed, packed=False):
    """``packed``: O comes back ONLY as a ``Packed`` bf16 operand (returns (Packed, probs)) - see ``attn_fwd_pack``.
    ``q`` may be the ``Packed`` fused Q | K | V projection (``k``, ``v`` None): the packed-input kernels, ``packed`` implied."""
    M = N * S
    in_pack = isinstance(q, Packed)
    dev = q.buf.device if in_pack else q.device
    packed = packed or in_pack
    if packed:
        o = None
        obuf = torch.empty((int(_lib.load().lstc_pack1_bytes(M, H * dv)),), device=dev, dtype=torch.uint8)
    else:
        o = torch.empty((M, H * dv), device=dev, dtype=torch.float32)
    if in_pack:      # rows padded to whole 16-B groups (include/lstc_hip.h, probs_ld); callers see the [N, H, S, S] view
        probs = torch.empty((N, H, S, (S + 3) // 4 * 4), device=dev, dtype=torch.float32)[..., :S]
    else:
        probs = torch.empty((N, H, S, S), device=dev, dtype=torch.float32)
    d = AttnDesc()
    d.N, d.S, d.H, d.dk, d.dv = N, S, H, dk, dv
    d.probs_ld = probs.stride(2)
    if in_pack:
        d.ldo = H * dv
    else:
        d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), (H * dv if packed else o.stride(0))
    d.dtype = _attn_dtype()
    if table is not None:
        d.index_ld, d.table_rows = index.shape[1], table.shape[0]
        d.table, d.index = dev_ptr(table), dev_ptr(index)
    d.scale = 1.0 / (dk ** 0.5)
    d.dropout_p, d.dropout_seed = float(p_drop), int(seed)
    d.variant = _ATTN_VARIANT
    d.probs = dev_ptr(probs)
    if in_pack:
        _qkv_pack_desc(d, q, H, dk, dv)
    else:
        d.Q, d.K, d.V = dev_ptr(q), dev_ptr(k), dev_ptr(v)
    if packed:
        d.O_pack = dev_ptr(obuf)
    else:
        d.O = dev_ptr(o)
    check(_lib.load().lstc_attn_fwd(C.byref(d), stream_ptr()), "lstc_attn_fwd")
    if packed:
        return Packed(obuf, M, H * dv, _lib.BF16P), probs
    return o, probs


def attn_bwd_packs(N, S, H, dk, dv) -> bool:
    """bf16 mode: the staged attention backward can write dQ / dK / dV as packed bf16 operands (include/lstc_hip.h)."""
    M = N * S
    return (_FUSE_PACKS and _packed_kind() == _lib.BF16P and _ATTN_VARIANT == 0 and S <= 96 and dk % 32 == 0 and dv % 32 == 0 and
            M % 256 == 0 and (H * dk) % 64 == 0 and (H * dv) % 64 == 0 and M * H * max(dk, dv) * 2 < 2 ** 31 and
            M >= max(_x3_min[0], 1) and H * min(dk, dv) >= max(_x3_min[1], 256) and M * H * min(dk, dv) * max(_x3_min[0], 256) >= _x3_min[2])


def attn_bwd(do, q, k, v, probs, N, S, H, dk, dv, table, index, p_drop, seed, out=None, packed=False):
    """``packed``: return dQ, dK, dV as ``Packed`` bf16 operands (no f32 copies) - see ``attn_bwd_packs``; ``packed="fused"``:
    ONE Packed [N*S, H*(2 dk + dv)] with the column blocks dQ | dK | dV (gradient of the fused Q|K|V projection).
    ``q`` and ``do`` may be ``Packed`` (the fused Q | K | V projection, the packed input gradient of fc; ``k``, ``v`` None): the
    packed-input kernel, ``packed="fused"`` implied."""
    in_pack = isinstance(q, Packed)
    if in_pack:
        if not (isinstance(do, Packed) and do.K == H * dv and do.kind == _lib.BF16P):
            raise RuntimeError("attn_bwd: a packed Q | K | V operand needs dO as a packed bf16 operand [N*S, H*d_v] too")
        packed = "fused"
    if packed:
        lib = _lib.load()
        M = N * S
        widths = (H * (2 * dk + dv),) if packed == "fused" else (H * dk, H * dk, H * dv)
        bufs = [torch.empty((int(lib.lstc_pack1_bytes(M, w)),), device=probs.device, dtype=torch.uint8) for w in widths]
        if packed == "fused":
            if M * widths[0] * 2 >= 2 ** 31:
                raise RuntimeError("attn_bwd(packed='fused'): pack larger than a buffer descriptor addresses")
            bufs = bufs * 3
        dq = dk_ = dv_ = None
    else:
        dq, dk_, dv_ = out if out is not None else (torch.empty_like(q), torch.empty_like(k), torch.empty_like(v))
    d = AttnDesc()
    d.N, d.S, d.H, d.dk, d.dv = N, S, H, dk, dv
    if not in_pack:
        d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), do.stride(0)
    if not packed and not (dq.stride(0) == q.stride(0) and dk_.stride(0) == k.stride(0) and dv_.stride(0) == v.stride(0)):
        raise RuntimeError("attn_bwd: dQ / dK / dV must have the row strides of Q / K / V")
    d.dtype = _attn_dtype()
    dtable = parts = None
    if table is not None:
        # one partial table per chunk of sequences, summed in a fixed order afterwards: no float atomics anywhere
        npw = min(8, max(1, (N * H + 4095) // 4096))
        if in_pack:        # the packed-input kernel's ring runs across a workgroup's sequences: fewer, longer workgroups (csrc/attention_pk.hip)
            npw = min(16, max(1, (N * H) // (8192 if S <= 32 else 1024)))
        if _BWD_NPW:
            npw = _BWD_NPW
        chunks = (N + npw - 1) // npw
        chunks = (N + ((N + chunks - 1) // chunks) - 1) // ((N + chunks - 1) // chunks)
        parts = torch.empty((chunks, table.shape[0] * H), device=probs.device, dtype=torch.float32)
        d.index_ld, d.table_rows, d.dtable_chunks = index.shape[1], table.shape[0], chunks
        d.table, d.index, d.dtable = dev_ptr(table), dev_ptr(index), dev_ptr(parts)
    d.scale = 1.0 / (dk ** 0.5)
    d.dropout_p, d.dropout_seed = float(p_drop), int(seed)
    d.probs, d.probs_ld = dev_ptr(probs), probs.stride(2)
    if not (probs.stride(3) == 1 and probs.stride(1) == S * probs.stride(2) and probs.stride(0) == H * S * probs.stride(2)):
        raise RuntimeError(f"attn_bwd: probs must be [N, H, S, S] rows of one pitch (strides {probs.stride()})")
    if in_pack:
        _qkv_pack_desc(d, q, H, dk, dv)
        d.dO, d.dO_pack_cols, d.dO_col0 = dev_ptr(do.buf), do.K, 0
    else:
        d.Q, d.K, d.V, d.dO = dev_ptr(q), dev_ptr(k), dev_ptr(v), dev_ptr(do)
    if packed:
        d.dQ_pack, d.dK_pack, d.dV_pack = (dev_ptr(b) for b in bufs)
        if packed == "fused":
            d.pack_cols, d.dQ_col0, d.dK_col0, d.dV_col0 = H * (2 * dk + dv), 0, H * dk, 2 * H * dk
    else:
        d.dQ, d.dK, d.dV = dev_ptr(dq), dev_ptr(dk_), dev_ptr(dv_)
    d.variant = _ATTN_VARIANT
    check(_lib.load().lstc_attn_bwd(C.byref(d), stream_ptr()), "lstc_attn_bwd")
    if parts is not None:
        dtable = colsum(parts).view(table.shape[0], H)
    if packed == "fused":
        return Packed(bufs[0], N * S, H * (2 * dk + dv), _lib.BF16P), None, None, dtable
    if packed:
        M = N * S
        return (Packed(bufs[0], M, H * dk, _lib.BF16P), Packed(bufs[1], M, H * dk, _lib.BF16P), Packed(bufs[2], M, H * dv, _lib.BF16P), dtable)
    return dq, dk_, dv_, dtable


def _fused_qkv_weight(wq, wk, wv):
    """[rows_q + rows_k + rows_v, d] view over the three projection weights when they are consecutive slices of one
    buffer (MultiHeadAttention.fuse_qkv_), else None.  The view object is kept on ``wq`` and reused while the buffer stays
    where it is: its packed copies are cached on it like a leaf weight's (``_packed_operand``) and rebuilt by
    ``repack_weights`` after an optimizer step."""
    try:
        same = wq.untyped_storage().data_ptr() == wk.untyped_storage().data_ptr() == wv.untyped_storage().data_ptr()
    except Exception:
        return None
    if not (same and wq.is_contiguous() and wk.is_contiguous() and wv.is_contiguous() and wq.shape[1] == wk.shape[1] == wv.shape[1]):
        return None
    if wk.storage_offset() != wq.storage_offset() + wq.numel() or wv.storage_offset() != wk.storage_offset() + wk.numel():
        return None
    rows = wq.shape[0] + wk.shape[0] + wv.shape[0]
    hit = wq.__dict__.get("_lstc_fused_view")
    if hit is not None and hit.data_ptr() == wq.data_ptr() and hit.shape == (rows, wq.shape[1]) and hit.device == wq.device:
        return hit
    view = torch.as_strided(wq.detach(), (rows, wq.shape[1]), (wq.shape[1], 1), wq.storage_offset())
    view._lstc_weight_view = True
    view.__dict__["_lstc_view_of"] = wq          # its packs go stale with wq's (one optimizer updates all three slices)
    wq.__dict__["_lstc_fused_view"] = view
    _weight_views.add(view)
    return view


_REPACK_WEIGHTS = os.environ.get("LSTC_NO_REPACK", "0") != "1"      # 0: weights are packed lazily, one launch per weight and layout (A/B)
_weight_views = weakref.WeakSet()       # fused-weight views whose packs ``repack_weights`` refreshes together with the leaf weights'


def repack_weights(params) -> int:
    """bf16 mode, after an optimizer step (which has just bumped the version of ``params``): rebuild in ONE launch
    (lstc_pack1_multi) every packed copy the finished step used of the given parameters and of THEIR fused Q|K|V views - into
    the same buffers, stamped with the new version, so the next step's products find them ready.  Only packs that were
    current before this step's bump are rebuilt; other optimizers' parameters are not touched.  Replaces the ~35 lazily issued lstc_pack1 launches of a step (10-20 us each, mostly
    ramp-up and drain: 0.45 ms of the LTN step, 0.76 ms of the STN step).  Returns the number of packs rebuilt."""
    if _packed_kind() != _lib.BF16P or not _REPACK_WEIGHTS:
        return 0
    items, stamp = [], []
    params = list(params)
    ids = {id(p) for p in params}
    views = [v for v in list(_weight_views) if id(v.__dict__.get("_lstc_view_of")) in ids]
    for t in params + views:
        cache = t.__dict__.get("_lstc_packs")
        if not cache:
            continue
        now = _wstamp(t)
        for key, hit in list(cache.items()):
            k_major, kind = key
            if kind != _lib.BF16P or hit[0] != (now[0], now[1] - 1) or hit[3] != t.data_ptr():
                continue
            src, pk = t.detach(), hit[2]
            if src.dim() != 2 or src.stride(1) != 1:
                continue
            r, c = src.shape
            rows, K = (c, r) if k_major else (r, c)
            if (pk.rows, pk.K) != (rows, K):
                continue
            items.append(_lib.PackItem(src.data_ptr(), rows, K, src.stride(0), int(k_major), pk.buf.data_ptr()))
            stamp.append((cache, key, t, pk))
    if not items:
        return 0
    arr = (_lib.PackItem * len(items))(*items)
    check(_lib.load().lstc_pack1_multi(arr, len(items), stream_ptr()), "lstc_pack1_multi")
    for cache, key, t, pk in stamp:
        cache[key] = (_wstamp(t), t._version, pk, t.data_ptr())
    return len(items)


# ------------------------------------------------------------------------------ autograd Functions
class MHAFunction(torch.autograd.Function):
    """models/MultiHeadAttention.py:93-132 (self-attention) as one autograd node.

    forward:  Q,K,V = X Wq^T, X Wk^T, X Wv^T -> fused attention core -> dropout(O Wfc^T) + X -> [LayerNorm]
    backward: hand-derived; every contraction is an lstc_gemm / lstc_attn_bwd launch."""

    @staticmethod
    def _forward_act(ctx, x, wq, wk, wv, wfc, ln_w, ln_b, table, index, cfg):
        """The block on the bf16 activation stream: ``x`` is the bf16 view of the pack of [N*S, d] (PackedAct.t).  Every product runs
        pack -> pack; the only f32 arrays are the probabilities and the LayerNorm statistics."""
        N, S, dm = cfg["act_shape"]
        M = N * S
        H, dk, dv = cfg["n_head"], cfg["d_k"], cfg["d_v"]
        training = cfg["training"]
        p_attn = cfg["attn_dropout"] if training else 0.0
        p_fc = cfg["fc_dropout"] if training else 0.0
        out16 = bool(cfg.get("act16_out", False))
        xp = _act_pack(x, M, dm)
        wqkv = _fused_qkv_weight(wq, wk, wv)
        if wqkv is None or not (attn_packed_inputs(N, S, H, dk, dv) and packed_out_shape(M, wqkv.shape[0]) and
                                packed_out_shape(M, H * dv) and packed_out_shape(M, dm)) or (not cfg["layer_norm"] and not out16):
            raise RuntimeError("MHAFunction: this layer / shape cannot run on the bf16 activation stream (functional.act_chain_ok)")
        qkv_p = gemm(xp, wqkv, trans_b=True, out_pack=True)
        seed_a = next_seed() if p_attn > 0 else 0
        seed_f = next_seed() if p_fc > 0 else 0
        if p_attn > 0:
            _note(cfg["site"] + "attn_dropout", p_attn, seed_a, (N, H, S, S))
        if p_fc > 0:
            _note(cfg["site"] + "dropout", p_fc, seed_f, (N, S, dm))
        op, probs = attn_fwd(qkv_p, None, None, N, S, H, dk, dv, table, index, p_attn, seed_a)
        yp = gemm(op, wfc, trans_b=True, dropout=(p_fc, seed_f), residual=xp, out_pack=True)
        if cfg["layer_norm"]:
            zf, zp, mean, rstd = layernorm_fwd_act(yp, ln_w, ln_b, 1e-6, want_f32=not out16, want_pack=out16)
        else:                    # no LayerNorm (models/MultiHeadAttention.py:125-126 skipped): the packed sum is the block's output
            zf, zp, mean, rstd = None, yp, None, None
        ctx.act = dict(xp=xp, qkv_p=qkv_p, op=op, yp=yp if cfg["layer_norm"] else None, out16=out16)
        ctx.cfg = dict(cfg, N=N, S=S, p_attn=p_attn, p_fc=p_fc, seed_a=seed_a, seed_f=seed_f)
        ctx.save_for_backward(wq, wk, wv, wfc, ln_w, table, index, probs, mean, rstd)
        ctx.mark_non_differentiable(probs)
        return (_act_tensor(zp) if out16 else zf.view(N, S, dm)), probs

    @staticmethod
    def _backward_act(ctx, dz):
        wq, wk, wv, wfc, ln_w, table, index, probs, mean, rstd = ctx.saved_tensors
        c, a = ctx.cfg, ctx.act
        N, S, H, dk, dv = c["N"], c["S"], c["n_head"], c["d_k"], c["d_v"]
        M = N * S
        xp, qkv_p, op, yp = a["xp"], a["qkv_p"], a["op"], a["yp"]
        dm = xp.K
        want_dx = ctx.needs_input_grad[0]
        dzin = _act_pack(dz.contiguous(), M, dm) if a["out16"] else dz.contiguous().view(M, dm)
        if c["layer_norm"]:
            dy, df, dln_w, dln_b, _ = layernorm_bwd_act(dzin, yp, ln_w, mean, rstd, c["p_fc"], c["seed_f"], want_dx, False)
        else:                    # z = dropout(f) + x: the incoming pack is the residual stream's gradient, its dropped form fc's
            dy, dln_w, dln_b = dzin, None, None
            df = dropout_apply_pack(dzin, c["p_fc"], c["seed_f"]) if c["p_fc"] > 0 else dzin
        dwfc = deliver(wfc, wgrad(df, None, op, out=grad_sink(wfc)))
        do = gemm(df, wfc, out_pack=True)
        wqkv = _fused_qkv_weight(wq, wk, wv)
        rq, rk = wq.shape[0], wk.shape[0]
        dqkv, _, _, dtable = attn_bwd(do, qkv_p, None, None, probs, N, S, H, dk, dv, table, index, c["p_attn"], c["seed_a"])
        dwqkv = wgrad(dqkv, None, xp, out=fused_grad_sink(wq, wk, wv))
        dwq, dwk, dwv = deliver(wq, dwqkv[:rq]), deliver(wk, dwqkv[rq: rq + rk]), deliver(wv, dwqkv[rq + rk:])
        dx = _act_tensor(gemm(dqkv, wqkv, residual=dy, out_pack=True)) if want_dx else None
        return dx, dwq, dwk, dwv, dwfc, dln_w, dln_b, dtable, None, None

    @staticmethod
    def forward(ctx, x, wq, wk, wv, wfc, ln_w, ln_b, table, index, cfg):
        ctx.is_act = x.dtype == torch.bfloat16
        if ctx.is_act:
            return MHAFunction._forward_act(ctx, x, wq, wk, wv, wfc, ln_w, ln_b, table, index, cfg)
        N, S, dm = x.shape
        H, dk, dv = cfg["n_head"], cfg["d_k"], cfg["d_v"]
        training = cfg["training"]
        p_attn = cfg["attn_dropout"] if training else 0.0
        p_fc = cfg["fc_dropout"] if training else 0.0
        x2 = x.contiguous().view(N * S, dm)
        xp = maybe_pack(x2)            # f32x3: one pack of X feeds Q, K, V now and the three weight gradients later
        xa = xp if xp is not None else x2
        wqkv = _fused_qkv_weight(wq, wk, wv)
        qkv_p = None
        if (xp is not None and wqkv is not None and wqkv.shape[0] == H * (2 * dk + dv) and attn_packed_inputs(N, S, H, dk, dv) and
                packed_out_shape(N * S, wqkv.shape[0]) and packed_out_shape(N * S, H * dv) and wfc.shape[0] >= max(_x3_min[0], 1)):
            # bf16 mode: Q | K | V exist only as ONE packed bf16 operand - the projection writes it (LSTC_EPI_OUT_PACK), the attention
            # core reads it (forward and backward) and writes O / dQ | dK | dV as packs: no f32 activation between the two GEMMs
            qkv_p = gemm(xa, wqkv, trans_b=True, out_pack=True)
            q = k = v = None
        elif wqkv is not None:       # w_qs / w_ks / w_vs live in one buffer (MultiHeadAttention.fuse_qkv_): one GEMM, X read once
            qkv = gemm(xa, wqkv, trans_b=True)
            q, k, v = qkv[:, : H * dk], qkv[:, H * dk: 2 * H * dk], qkv[:, 2 * H * dk:]
        else:
            q = gemm(xa, wq, trans_b=True)
            k = gemm(xa, wk, trans_b=True)
            v = gemm(xa, wv, trans_b=True)
        seed_a = next_seed() if p_attn > 0 else 0
        seed_f = next_seed() if p_fc > 0 else 0
        if p_attn > 0:
            _note(cfg["site"] + "attn_dropout", p_attn, seed_a, (N, H, S, S))
        if p_fc > 0:
            _note(cfg["site"] + "dropout", p_fc, seed_f, (N, S, dm))
        if qkv_p is not None:
            op, probs = attn_fwd(qkv_p, None, None, N, S, H, dk, dv, table, index, p_attn, seed_a)
            o = None
        elif xp is not None and attn_fwd_pack(N, S, H, dv) and wfc.shape[0] >= max(_x3_min[0], 1):
            # bf16 mode: the attention output exists only as the packed bf16 operand of fc (and of fc's weight gradient)
            op, probs = attn_fwd(q, k, v, N, S, H, dk, dv, table, index, p_attn, seed_a, packed=True)
            o = None
        else:
            o, probs = attn_fwd(q, k, v, N, S, H, dk, dv, table, index, p_attn, seed_a)
            op = maybe_pack(o)
        y = gemm(op if op is not None else o, wfc, trans_b=True, dropout=(p_fc, seed_f), residual=x2)
        if cfg["layer_norm"]:
            z, mean, rstd = layernorm_fwd(y, ln_w, ln_b, 1e-6, pack=True)
        else:
            z, mean, rstd = y, None, None
        ctx.packs = (xp, op) if (training or o is None) else (None, None)
        ctx.qkv_p = qkv_p
        ctx.cfg = dict(cfg, N=N, S=S, p_attn=p_attn, p_fc=p_fc, seed_a=seed_a, seed_f=seed_f)
        ctx.save_for_backward(x2, wq, wk, wv, wfc, ln_w, table, index, q, k, v, o, probs,
                              y if cfg["layer_norm"] else None, mean, rstd)
        ctx.mark_non_differentiable(probs)
        return z.view(N, S, dm), probs

    @staticmethod
    def backward(ctx, dz, _dprobs):
        if ctx.is_act:
            return MHAFunction._backward_act(ctx, dz)
        x2, wq, wk, wv, wfc, ln_w, table, index, q, k, v, o, probs, y, mean, rstd = ctx.saved_tensors
        c = ctx.cfg
        N, S, H, dk, dv = c["N"], c["S"], c["n_head"], c["d_k"], c["d_v"]
        dz2 = dz.contiguous().view(N * S, -1)
        dy, df, dln_w, dln_b, _ = layernorm_bwd_branch(dz2, y, ln_w, mean, rstd, c["p_fc"], c["seed_f"], c["layer_norm"], False)
        xp, op = ctx.packs
        dwfc = deliver(wfc, wgrad(df, o, op, out=grad_sink(wfc)))
        qkv_p = ctx.qkv_p
        do = gemm(df, wfc, out_pack=qkv_p is not None)       # [M, H*dv]; packed-input attention: as a packed bf16 operand only
        wqkv = _fused_qkv_weight(wq, wk, wv)
        dx = None
        if wqkv is not None:
            rq, rk = wq.shape[0], wk.shape[0]
            if qkv_p is not None:
                dqkv, _, _, dtable = attn_bwd(do, qkv_p, None, None, probs, N, S, H, dk, dv, table, index, c["p_attn"], c["seed_a"])
            elif xp is not None and attn_bwd_packs(N, S, H, dk, dv) and N * S * wqkv.shape[0] * 2 < 2 ** 31:
                # bf16 mode: the attention backward writes dQ | dK | dV straight into ONE packed bf16 operand
                dqkv, _, _, dtable = attn_bwd(do, q, k, v, probs, N, S, H, dk, dv, table, index, c["p_attn"], c["seed_a"], packed="fused")
            else:
                dqkv = torch.empty((N * S, wqkv.shape[0]), device=x2.device, dtype=torch.float32)
                outs = (dqkv[:, :rq], dqkv[:, rq: rq + rk], dqkv[:, rq + rk:])
                _, _, _, dtable = attn_bwd(do, q, k, v, probs, N, S, H, dk, dv, table, index, c["p_attn"], c["seed_a"], out=outs)
            dwqkv = wgrad(dqkv, x2, xp, out=fused_grad_sink(wq, wk, wv))      # one TN GEMM for the three weight gradients
            dwq, dwk, dwv = deliver(wq, dwqkv[:rq]), deliver(wk, dwqkv[rq: rq + rk]), deliver(wv, dwqkv[rq + rk:])
            if ctx.needs_input_grad[0]:
                dx = gemm(dqkv, wqkv, residual=dy).view(N, S, -1)   # dQ Wq + dK Wk + dV Wv + residual in one GEMM (K = 3*H*dk)
        else:
            dq, dk_, dv_, dtable = attn_bwd(do, q, k, v, probs, N, S, H, dk, dv, table, index, c["p_attn"], c["seed_a"],
                                            packed=xp is not None and attn_bwd_packs(N, S, H, dk, dv))
            dwq = deliver(wq, wgrad(dq, x2, xp, out=grad_sink(wq)))
            dwk = deliver(wk, wgrad(dk_, x2, xp, out=grad_sink(wk)))
            dwv = deliver(wv, wgrad(dv_, x2, xp, out=grad_sink(wv)))
            if ctx.needs_input_grad[0]:
                dx = gemm(dq, wq, residual=dy)
                gemm(dk_, wk, out=dx, accumulate=True)
                gemm(dv_, wv, out=dx, accumulate=True)
                dx = dx.view(N, S, -1)
        return dx, dwq, dwk, dwv, dwfc, dln_w, dln_b, dtable, None, None


def gemm_batched(a, b, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, batch, sa, sb, sc, alpha=1.0, a_off=0, b_off=0, c_off=0):
    """``batch`` independent GEMMs of one shape in one launch (grid.z): problem z uses a + z*sa, b + z*sb, c + z*sc
    (element strides).  ``a``/``b``/``c`` are the underlying tensors; leading dimensions are given explicitly because
    the per-head operands are column slices of wider matrices."""
    d = GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, lda, ldb, ldc
    d.transA, d.transB, d.flags, d.alpha = int(trans_a), int(trans_b), 0, float(alpha)
    d.dtype = F32 if _compute_dtype == _lib.F32X3 else _compute_dtype          # the packed kernel has no batch mode
    d.batch, d.batch_stride_a, d.batch_stride_b, d.batch_stride_c = batch, sa, sb, sc
    d.A, d.B, d.C = dev_ptr(a) + 4 * a_off, dev_ptr(b) + 4 * b_off, dev_ptr(c) + 4 * c_off
    if _gemm_prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.load().lstc_gemm(C.byref(d), stream_ptr()), "lstc_gemm(batched)")
        e1.record()
        _gemm_prof.append((2.0 * M * N * K * batch, e0, e1))
        return c
    check(_lib.load().lstc_gemm(C.byref(d), stream_ptr()), "lstc_gemm(batched)")
    return c


def cls_dot(u, x3, mode, probs=None, p_drop=0.0, seed=0):
    N, S, dm = x3.shape
    H = u.shape[1]
    out = torch.empty((N, H, S), device=u.device, dtype=torch.float32)
    if mode == 1:
        probs = torch.empty((N, H, S), device=u.device, dtype=torch.float32)
    check(_lib.load().lstc_cls_dot(dev_ptr(u), dev_ptr(x3), dev_ptr(out), dev_ptr(probs), N, S, H, dm, mode, float(p_drop),
                                   int(seed), stream_ptr()), "lstc_cls_dot")
    return out, probs


def cls_wsum(w, x3):
    N, S, dm = x3.shape
    H = w.shape[1]
    y = torch.empty((N, H, dm), device=w.device, dtype=torch.float32)
    check(_lib.load().lstc_cls_wsum(dev_ptr(w), dev_ptr(x3), dev_ptr(y), N, S, H, dm, stream_ptr()), "lstc_cls_wsum")
    return y


def cls_outer(w1, u1, w2, u2, N, S, dm):
    H = w1.shape[1]
    dx = torch.empty((N, S, dm), device=w1.device, dtype=torch.float32)
    check(_lib.load().lstc_cls_outer(dev_ptr(w1), dev_ptr(u1), dev_ptr(w2), dev_ptr(u2), dev_ptr(dx), N, S, H, dm,
                                     stream_ptr()), "lstc_cls_outer")
    return dx


def unpack1_rows(pk: "Packed", row0: int = 0, step: int = 1, n: Optional[int] = None) -> torch.Tensor:
    """Rows ``row0 + i * step`` (i < n) of a packed bf16 operand as f32 [n, K] (lstc_unpack1_rows)."""
    if pk.kind != _lib.BF16P:
        raise TypeError("unpack1_rows reads lstc_pack1 operands")
    n = (pk.rows - row0 + step - 1) // step if n is None else n
    out = torch.empty((n, pk.K), device=pk.buf.device, dtype=torch.float32)
    check(_lib.load().lstc_unpack1_rows(dev_ptr(pk.buf), pk.rows, pk.K, row0, step, n, dev_ptr(out), pk.K, stream_ptr()),
          "lstc_unpack1_rows")
    return out


def cls_pack_ok(N: int, S: int, H: int, dm: int) -> bool:
    """Can the CLS-only layer read its [N, S, dm] input as a pack (include/lstc_hip.h, lstc_cls_dot_pack)?"""
    return _ACT16 and _CLS_PACK and H <= 8 and S <= 128 and act_rows_ok(N * S, dm)


def cls_dot_pack(u, xp: "Packed", N, S, mode, probs=None, p_drop=0.0, seed=0):
    H = u.shape[1]
    out = torch.empty((N, H, S), device=u.device, dtype=torch.float32)
    if mode == 1:
        probs = torch.empty((N, H, S), device=u.device, dtype=torch.float32)
    check(_lib.load().lstc_cls_dot_pack(dev_ptr(u), dev_ptr(xp.buf), dev_ptr(out), dev_ptr(probs), N, S, H, xp.K, mode, float(p_drop),
                                        int(seed), stream_ptr()), "lstc_cls_dot_pack")
    return out, probs


def cls_wsum_pack(w, xp: "Packed", N, S):
    H = w.shape[1]
    y = torch.empty((N, H, xp.K), device=w.device, dtype=torch.float32)
    check(_lib.load().lstc_cls_wsum_pack(dev_ptr(w), dev_ptr(xp.buf), dev_ptr(y), N, S, H, xp.K, stream_ptr()), "lstc_cls_wsum_pack")
    return y


def cls_outer_pack(w1, u1, w2, u2, add0, N, S, dm) -> "Packed":
    H = w1.shape[1]
    buf = torch.empty((int(_lib.load().lstc_pack1_bytes(N * S, dm)),), device=w1.device, dtype=torch.uint8)
    check(_lib.load().lstc_cls_outer_pack(dev_ptr(w1), dev_ptr(u1), dev_ptr(w2), dev_ptr(u2), dev_ptr(add0), dev_ptr(buf), N, S, H, dm,
                                          stream_ptr()), "lstc_cls_outer_pack")
    return Packed(buf, N * S, dm, _lib.BF16P)


class MHAClsAssocFunction(torch.autograd.Function):
    """Last-layer CLS attention with the key / value projections re-associated away (csrc/attention.hip, "assoc"):
        u[n,h]  = (q[n,h] / sqrt(dk)) Wk_h          score[n,h,j] = u[n,h] . x[n,j]        (no K = X Wk^T GEMM)
        xb[n,h] = sum_j Pd[n,h,j] x[n,j]            o[n,h]       = Wv_h xb[n,h]           (no V = X Wv^T GEMM)
    Exactly models/MultiHeadAttention.py:93-126 for query row 0 — same values up to float re-association — but the two
    [N*S, d] x [d, H*dk] projections and their four backward GEMMs (5 of the 43 TFLOP of an LTN step) become per-head
    [N, dk] x [dk, d] products (one batched launch each) plus three passes over X."""

    @staticmethod
    @_small_m_fwd
    def forward(ctx, x, wq, wk, wv, wfc, ln_w, ln_b, table, cfg):
        # bf16 activation stream: x is the bf16 view of the pack of the last full layer's output (cfg["act_shape"]); the three
        # passes over X read the pack (half the bytes), the CLS rows come out of it widened
        xp = None
        if x.dtype == torch.bfloat16:
            N, S, dm = cfg["act_shape"]
            xp = _act_pack(x, N * S, dm)
        else:
            N, S, dm = x.shape
        H, dk, dv = cfg["n_head"], cfg["d_k"], cfg["d_v"]
        training = cfg["training"]
        p_attn = cfg["attn_dropout"] if training else 0.0
        p_fc = cfg["fc_dropout"] if training else 0.0
        ctx.table_shape = None if table is None else tuple(table.shape)
        if xp is None:
            x = x.contiguous()
            xc = x[:, 0, :]
        else:
            xc = unpack1_rows(xp, 0, S, N)
        scale = 1.0 / (dk ** 0.5)
        qc = gemm(xc, wq, trans_b=True)                                                   # [N, H*dk]
        u = torch.empty((N, H, dm), device=x.device, dtype=torch.float32)
        gemm_batched(qc, wk, u, N, dm, dk, H * dk, dm, H * dm, False, False, H, dk, dk * dm, dm, alpha=scale)
        seed_a = next_seed() if p_attn > 0 else 0
        seed_f = next_seed() if p_fc > 0 else 0
        if p_attn > 0:
            _note(cfg["site"] + "attn_dropout#cls", p_attn, seed_a, (N, H, S, S))
        if p_fc > 0:
            _note(cfg["site"] + "dropout#cls", p_fc, seed_f, (N, dm))
        if xp is None:
            pd, probs = cls_dot(u, x, 1, None, p_attn, seed_a)                            # dropped probs, probs
            xb = cls_wsum(pd, x)                                                          # [N, H, d]
        else:
            pd, probs = cls_dot_pack(u, xp, N, S, 1, None, p_attn, seed_a)
            xb = cls_wsum_pack(pd, xp, N, S)
        oc = torch.empty((N, H * dv), device=x.device, dtype=torch.float32)
        per_head_rows_wT(xb, wv, oc, N, dv, dm, H)
        y = gemm(oc, wfc, trans_b=True, dropout=(p_fc, seed_f), residual=xc)
        if cfg["layer_norm"]:
            z, mean, rstd = layernorm_fwd(y, ln_w, ln_b, 1e-6)
        else:
            z, mean, rstd = y, None, None
        ctx.cfg = dict(cfg, N=N, S=S, dm=dm, is_act=xp is not None, p_attn=p_attn, p_fc=p_fc, seed_a=seed_a, seed_f=seed_f, scale=scale)
        ctx.save_for_backward(x, wq, wk, wv, wfc, ln_w, qc, u, pd, probs, xb, oc, y if cfg["layer_norm"] else None, mean, rstd,
                              xc if xp is not None else None)
        return z

    @staticmethod
    @_small_m_bwd
    def backward(ctx, dz):
        x, wq, wk, wv, wfc, ln_w, qc, u, pd, probs, xb, oc, y, mean, rstd, xc_saved = ctx.saved_tensors
        c = ctx.cfg
        N, S, H, dk, dv, scale = c["N"], c["S"], c["n_head"], c["d_k"], c["d_v"], c["scale"]
        dm = c["dm"]
        xp = _act_pack(x, N * S, dm) if c["is_act"] else None
        xc = xc_saved if xp is not None else x[:, 0, :]
        dz = dz.contiguous()
        dln_w = dln_b = None
        if c["layer_norm"]:
            dy, dln_w, dln_b = layernorm_bwd(dz, y, ln_w, mean, rstd)
        else:
            dy = dz
        df = dropout_apply(dy, c["p_fc"], c["seed_f"]) if c["p_fc"] > 0 else dy
        dwfc = deliver(wfc, wgrad(df, oc, out=grad_sink(wfc)))
        doc = gemm(df, wfc)                                                               # [N, H*dv]
        dxb = torch.empty((N, H, dm), device=x.device, dtype=torch.float32)
        gemm_batched(doc, wv, dxb, N, dm, dv, H * dv, dm, H * dm, False, False, H, dv, dv * dm, dm)
        dwv = grad_sink(wv)
        dwv = torch.empty_like(wv) if dwv is None else dwv
        gemm_batched(doc, xb, dwv, dv, dm, N, H * dv, H * dm, dm, True, False, H, dv, dm, dv * dm)
        if xp is None:
            ds, _ = cls_dot(dxb, x, 2, probs, c["p_attn"], c["seed_a"])                   # d(logit) [N,H,S]
            du = cls_wsum(ds, x)                                                          # [N, H, d]
        else:
            ds, _ = cls_dot_pack(dxb, xp, N, S, 2, probs, c["p_attn"], c["seed_a"])
            du = cls_wsum_pack(ds, xp, N, S)
        dqc = torch.empty_like(qc)
        per_head_rows_wT(du, wk, dqc, N, dk, dm, H, alpha=scale)
        dwk = grad_sink(wk)
        dwk = torch.empty_like(wk) if dwk is None else dwk
        gemm_batched(qc, du, dwk, dk, dm, N, H * dk, H * dm, dm, True, False, H, dk, dm, dk * dm, alpha=scale)
        dwq = wgrad(dqc, xc, out=grad_sink(wq))
        dx = None
        if ctx.needs_input_grad[0]:
            if xp is None:
                dx = cls_outer(pd, dxb, ds, u, N, S, dm)                                  # K/V paths, every token
                gemm(dqc, wq, out=dx[:, 0, :], accumulate=True, residual=dy)              # CLS rows: + dQ Wq + residual
            else:       # the gradient goes back as a pack; the CLS rows' own terms join row 0 before the one rounding
                dx = _act_tensor(cls_outer_pack(pd, dxb, ds, u, gemm(dqc, wq, residual=dy), N, S, dm))
        dtable = None if ctx.table_shape is None else torch.zeros(ctx.table_shape, device=x.device, dtype=torch.float32)
        return dx, deliver(wq, dwq), deliver(wk, dwk), deliver(wv, dwv), dwfc, dln_w, dln_b, dtable, None


def attn_cls_fwd(qc, k, v, N, S, H, dk, dv, p_drop, seed):
    oc = torch.empty((N, H * dv), device=qc.device, dtype=torch.float32)
    probs = torch.empty((N, H, S), device=qc.device, dtype=torch.float32)
    d = AttnDesc()
    d.N, d.S, d.H, d.dk, d.dv = N, S, H, dk, dv
    d.ldq, d.ldk, d.ldv, d.ldo = qc.stride(0), k.stride(0), v.stride(0), oc.stride(0)
    d.dtype, d.scale = F32, 1.0 / (dk ** 0.5)
    d.dropout_p, d.dropout_seed = float(p_drop), int(seed)
    d.Q, d.K, d.V, d.O, d.probs = dev_ptr(qc), dev_ptr(k), dev_ptr(v), dev_ptr(oc), dev_ptr(probs)
    check(_lib.load().lstc_attn_cls_fwd(C.byref(d), stream_ptr()), "lstc_attn_cls_fwd")
    return oc, probs


def attn_cls_bwd(doc, qc, k, v, probs, N, S, H, dk, dv, p_drop, seed):
    dqc, dk_, dv_ = torch.empty_like(qc), torch.empty_like(k), torch.empty_like(v)
    d = AttnDesc()
    d.N, d.S, d.H, d.dk, d.dv = N, S, H, dk, dv
    d.ldq, d.ldk, d.ldv, d.ldo = qc.stride(0), k.stride(0), v.stride(0), doc.stride(0)
    d.dtype, d.scale = F32, 1.0 / (dk ** 0.5)
    d.dropout_p, d.dropout_seed = float(p_drop), int(seed)
    d.Q, d.K, d.V, d.probs = dev_ptr(qc), dev_ptr(k), dev_ptr(v), dev_ptr(probs)
    d.dO, d.dQ, d.dK, d.dV = dev_ptr(doc), dev_ptr(dqc), dev_ptr(dk_), dev_ptr(dv_)
    check(_lib.load().lstc_attn_cls_bwd(C.byref(d), stream_ptr()), "lstc_attn_cls_bwd")
    return dqc, dk_, dv_


class MHAClsFunction(torch.autograd.Function):
    """Self-attention of the LAST encoder layer restricted to the CLS query: the train loops read
    only ``enc_output[:, 0, :]`` (Train/temporal_transformer_shanghaitech.py:123), so Q, the output projection,
    dropout, residual and LayerNorm are evaluated for token 0 only; K and V still cover every token.
    x [N, S, d] -> [N, d].  Same math as MHAFunction row 0 (models/MultiHeadAttention.py:93-126)."""

    @staticmethod
    @_small_m_fwd
    def forward(ctx, x, wq, wk, wv, wfc, ln_w, ln_b, table, cfg):
        # `table`: the layer's relative-position bias table.  Row 0 never sees the bias, so its gradient is exactly
        # zero — but the reference still hands Adagrad that zero tensor, and Adagrad applies weight decay to it
        # (grad != None).  backward() therefore returns zeros for it instead of None to keep the update identical.
        N, S, dm = x.shape
        H, dk, dv = cfg["n_head"], cfg["d_k"], cfg["d_v"]
        training = cfg["training"]
        p_attn = cfg["attn_dropout"] if training else 0.0
        p_fc = cfg["fc_dropout"] if training else 0.0
        ctx.table_shape = None if table is None else tuple(table.shape)
        x = x.contiguous()
        x2 = x.view(N * S, dm)
        xc = x[:, 0, :]                                     # [N, d] view, row stride S*d
        k = gemm(x2, wk, trans_b=True)
        v = gemm(x2, wv, trans_b=True)
        qc = gemm(xc, wq, trans_b=True)
        seed_a = next_seed() if p_attn > 0 else 0
        seed_f = next_seed() if p_fc > 0 else 0
        if p_attn > 0:
            _note(cfg["site"] + "attn_dropout#cls", p_attn, seed_a, (N, H, S, S))     # row 0 of the full mask
        if p_fc > 0:
            _note(cfg["site"] + "dropout#cls", p_fc, seed_f, (N, dm))
        oc, probs = attn_cls_fwd(qc, k, v, N, S, H, dk, dv, p_attn, seed_a)
        y = gemm(oc, wfc, trans_b=True, dropout=(p_fc, seed_f), residual=xc)
        if cfg["layer_norm"]:
            z, mean, rstd = layernorm_fwd(y, ln_w, ln_b, 1e-6)
        else:
            z, mean, rstd = y, None, None
        ctx.cfg = dict(cfg, N=N, S=S, p_attn=p_attn, p_fc=p_fc, seed_a=seed_a, seed_f=seed_f)
        ctx.save_for_backward(x, wq, wk, wv, wfc, ln_w, qc, k, v, oc, probs, y if cfg["layer_norm"] else None, mean, rstd)
        return z

    @staticmethod
    @_small_m_bwd
    def backward(ctx, dz):
        x, wq, wk, wv, wfc, ln_w, qc, k, v, oc, probs, y, mean, rstd = ctx.saved_tensors
        c = ctx.cfg
        N, S, H, dk, dv = c["N"], c["S"], c["n_head"], c["d_k"], c["d_v"]
        dm = x.shape[-1]
        x2, xc = x.view(N * S, dm), x[:, 0, :]
        dz = dz.contiguous()
        dln_w = dln_b = None
        if c["layer_norm"]:
            dy, dln_w, dln_b = layernorm_bwd(dz, y, ln_w, mean, rstd)
        else:
            dy = dz
        df = dropout_apply(dy, c["p_fc"], c["seed_f"]) if c["p_fc"] > 0 else dy
        dwfc = deliver(wfc, wgrad(df, oc, out=grad_sink(wfc)))
        doc = gemm(df, wfc)
        dqc, dk_, dv_ = attn_cls_bwd(doc, qc, k, v, probs, N, S, H, dk, dv, c["p_attn"], c["seed_a"])
        dwq = deliver(wq, wgrad(dqc, xc, out=grad_sink(wq)))
        dwk, dwv = deliver(wk, wgrad(dk_, x2, out=grad_sink(wk))), deliver(wv, wgrad(dv_, x2, out=grad_sink(wv)))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dk_, wk)
            gemm(dv_, wv, out=dx, accumulate=True)
            gemm(dqc, wq, out=dx.view(N, S, dm)[:, 0, :], accumulate=True, residual=dy)   # CLS rows: + dQ Wq + residual
            dx = dx.view(N, S, dm)
        dtable = None if ctx.table_shape is None else torch.zeros(ctx.table_shape, device=x.device, dtype=torch.float32)
        return dx, dwq, dwk, dwv, dwfc, dln_w, dln_b, dtable, None


_PAD_HIDDEN = os.environ.get("LSTC_NO_PAD_HIDDEN", "0") != "1"


def _padded_hidden(F: int) -> int:
    """Width the FFN block runs at for n_hidden = F: the next multiple of 256 (full GEMM tiles in every mode, packed bf16
    hidden) when that adds at most 5 % of columns, else the next multiple of 4 (16-B aligned rows); F itself when aligned."""
    if not _PAD_HIDDEN or F % 256 == 0:
        return F
    up = (F + 255) // 256 * 256
    if (up - F) * 20 <= F:
        return up
    return (F + 3) // 4 * 4


def _ffn_padded_weights(w1, b1, w2, Fp, device):
    """W1, b1, W2 at the padded hidden width ``Fp`` (zero rows / bias entries / columns appended; copies cached on the W1 parameter
    until the weights change: evaluation loops call the block per video)."""
    F, dm = w1.shape
    stamp = (_wstamp(w1), _wstamp(w2), _wstamp(b1), Fp, w1._version, b1._version, w2._version, w1.data_ptr(), b1.data_ptr(), w2.data_ptr())
    hit = w1.__dict__.get("_lstc_padded")
    if hit is None or hit[0] != stamp:
        w1p = torch.zeros((Fp, dm), device=device, dtype=torch.float32)
        w1p[:F].copy_(w1.detach())
        b1p = torch.zeros((Fp,), device=device, dtype=torch.float32)
        b1p[:F].copy_(b1.detach())
        w2p = torch.zeros((w2.shape[0], Fp), device=device, dtype=torch.float32)
        w2p[:, :F].copy_(w2.detach())
        hit = (stamp, (w1p, b1p, w2p))
        w1.__dict__["_lstc_padded"] = hit
    return hit[1]


class FFNFunction(torch.autograd.Function):
    """models/FFN.py:14-22: LN?( dropout(W2 relu(W1 x + b1) + b2) + x )."""

    @staticmethod
    def _forward_act(ctx, x, w1, b1, w2, b2, ln_w, ln_b, cfg):
        """The block on the bf16 activation stream (see MHAFunction._forward_act)."""
        N, S, dm = cfg["act_shape"]
        M = N * S
        p = cfg["dropout"] if cfg["training"] else 0.0
        seed = next_seed() if p > 0 else 0
        if p > 0:
            _note(cfg["site"] + "dropout", p, seed, (N, S, dm))
        out16 = bool(cfg.get("act16_out", False))
        F = w1.shape[0]
        Fp = _padded_hidden(F)
        ctx.F = F
        ctx.w_params = (w1, w2)
        if Fp != F:
            w1, b1, w2 = _ffn_padded_weights(w1, b1, w2, Fp, x.device)
        if not cfg["layer_norm"] or not (packed_out_shape(M, Fp) and packed_out_shape(M, dm)):
            raise RuntimeError("FFNFunction: this layer / shape cannot run on the bf16 activation stream (functional.act_chain_ok)")
        xp = _act_pack(x, M, dm)
        hp = gemm(xp, w1, trans_b=True, bias=b1, relu=True, out_pack=True)
        yp = gemm(hp, w2, trans_b=True, bias=b2, dropout=(p, seed), residual=xp, out_pack=True)
        zf, zp, mean, rstd = layernorm_fwd_act(yp, ln_w, ln_b, 1e-6, want_f32=not out16, want_pack=out16)
        ctx.act = dict(xp=xp, hp=hp, yp=yp, out16=out16)
        ctx.cfg = dict(cfg, p=p, seed=seed, shape=(N, S, dm))
        ctx.save_for_backward(w1, w2, ln_w, mean, rstd)
        return _act_tensor(zp) if out16 else zf.view(N, S, dm)

    @staticmethod
    def _backward_act(ctx, dz):
        w1, w2, ln_w, mean, rstd = ctx.saved_tensors
        c, a = ctx.cfg, ctx.act
        N, S, dm = c["shape"]
        M = N * S
        xp, hp, yp = a["xp"], a["hp"], a["yp"]
        want_dx = ctx.needs_input_grad[0]
        dzin = _act_pack(dz.contiguous(), M, dm) if a["out16"] else dz.contiguous().view(M, dm)
        dy, df, dln_w, dln_b, db2 = layernorm_bwd_act(dzin, yp, ln_w, mean, rstd, c["p"], c["seed"], want_dx, True)
        w1o, w2o = ctx.w_params
        padded = w1.shape[0] != ctx.F
        s1, s2 = (None, None) if padded else (grad_sink(w1o), grad_sink(w2o))
        dw2 = wgrad(df, None, hp, out=s2)
        dh1 = gemm(df, w2, relu_mask=hp, out_pack=True)
        db1 = colsum_pack(dh1)
        dw1 = wgrad(dh1, None, xp, out=s1)
        dx = _act_tensor(gemm(dh1, w1, residual=dy, out_pack=True)) if want_dx else None
        if padded:
            dw1, db1, dw2 = dw1[:ctx.F], db1[:ctx.F].contiguous(), dw2[:, :ctx.F]
            if grad_sink(w2o) is None:
                dw2 = dw2.contiguous()
        return dx, deliver(w1o, dw1), db1, deliver(w2o, dw2), db2, dln_w, dln_b, None

    @staticmethod
    @_small_m_fwd
    def forward(ctx, x, w1, b1, w2, b2, ln_w, ln_b, cfg):
        ctx.is_act = x.dtype == torch.bfloat16
        if ctx.is_act:
            return FFNFunction._forward_act(ctx, x, w1, b1, w2, b2, ln_w, ln_b, cfg)
        shape = x.shape
        dm = shape[-1]
        x2 = x.contiguous().view(-1, dm)
        p = cfg["dropout"] if cfg["training"] else 0.0
        seed = next_seed() if p > 0 else 0
        if p > 0:
            _note(cfg["site"] + "dropout", p, seed, shape)
        F = w1.shape[0]
        Fp = _padded_hidden(F)
        ctx.F = F
        ctx.w_params = (w1, w2)          # the Parameters themselves (below they may be replaced by padded copies): gradient sinks hang on them
        if Fp != F:
            # hidden width that leaves rows unaligned (the reference's STN: n_hidden = 3027): run the block at the padded width with
            # zero rows / columns / bias entries appended to W1, b1, W2 - the extra hidden units are relu(0) = 0 and meet zero
            # weights, so y, dx and the real rows of every gradient are unchanged, and all five products with the hidden in
            # them take the aligned (vector-load, full-tile, packable) paths
            w1, b1, w2 = _ffn_padded_weights(w1, b1, w2, Fp, x2.device)
        xp = maybe_pack(x2)
        if xp is not None and packed_out_shape(x2.shape[0], w1.shape[0]) and w2.shape[0] >= max(_x3_min[0], 1):
            # bf16 mode: the hidden exists only as the packed bf16 operand W2 (and dW2, and the ReLU mask of the backward) reads
            h1, hp = None, gemm(xp, w1, trans_b=True, bias=b1, relu=True, out_pack=True)
        else:
            h1 = gemm(xp if xp is not None else x2, w1, trans_b=True, bias=b1, relu=True)
            hp = maybe_pack(h1)
        y = gemm(hp if hp is not None else h1, w2, trans_b=True, bias=b2, dropout=(p, seed), residual=x2)
        ctx.packs = (xp, hp) if (cfg["training"] or h1 is None) else (None, None)
        if cfg["layer_norm"]:
            # the packed bf16 copy of the result is worth its 2 B / element only when a full encoder layer consumes it next
            # (Encoder sets emit_pack: not for the layer that feeds the CLS-only last layer or the caller)
            z, mean, rstd = layernorm_fwd(y, ln_w, ln_b, 1e-6, pack=cfg.get("emit_pack", True))
        else:
            z, mean, rstd = y, None, None
        ctx.cfg = dict(cfg, p=p, seed=seed, shape=tuple(shape))
        ctx.save_for_backward(x2, w1, w2, ln_w, h1, y if cfg["layer_norm"] else None, mean, rstd)
        return z.view(shape)

    @staticmethod
    @_small_m_bwd
    def backward(ctx, dz):
        if ctx.is_act:
            return FFNFunction._backward_act(ctx, dz)
        x2, w1, w2, ln_w, h1, y, mean, rstd = ctx.saved_tensors
        c = ctx.cfg
        dz2 = dz.contiguous().view(-1, dz.shape[-1])
        dy, df, dln_w, dln_b, db2 = layernorm_bwd_branch(dz2, y, ln_w, mean, rstd, c["p"], c["seed"], c["layer_norm"], True)
        xp, hp = ctx.packs
        w1o, w2o = ctx.w_params
        padded = w1.shape[0] != ctx.F
        s1, s2 = (None, None) if padded else (grad_sink(w1o), grad_sink(w2o))      # padded width: computed wide, the real part copied out
        if h1 is None:        # packed hidden (forward): its gradient also lives only in packed form
            dw2 = wgrad(df, None, hp, out=s2)
            dh1 = gemm(df, w2, relu_mask=hp, out_pack=True)                  # [M, F] packed, relu' from the packed hidden's sign
            db1 = colsum_pack(dh1)
        else:
            dw2 = wgrad(df, h1, hp, out=s2)
            dh1 = gemm(df, w2, relu_mask=h1)                     # [M, F], relu' fused
            db1 = colsum(dh1)
        dw1 = wgrad(dh1, x2, xp, out=s1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dh1, w1, residual=dy).view(c["shape"])
        if padded:                        # padded hidden width: the real rows / columns of the gradients
            dw1, db1, dw2 = dw1[:ctx.F], db1[:ctx.F].contiguous(), dw2[:, :ctx.F]
            if grad_sink(w2o) is None:
                dw2 = dw2.contiguous()
        return dx, deliver(w1o, dw1), db1, deliver(w2o, dw2), db2, dln_w, dln_b, None


class LayerNormFunction(torch.autograd.Function):
    """nn.LayerNorm(d_model, eps=1e-6) on the Encoder input (models/Encoder.py:48-49)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.contiguous().view(-1, x.shape[-1])
        y, mean, rstd = layernorm_fwd(x2, w, b, 1e-6)
        ctx.save_for_backward(x2, w, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = layernorm_bwd(dy.contiguous().view(x2.shape), x2, w, mean, rstd)
        return (dx.view(dy.shape) if ctx.needs_input_grad[0] else None), dw, db


class ClsConcatFunction(torch.autograd.Function):
    """models/Encoder.py:51-58: CLS (token mean or learned) prepended, optional learned position table added."""

    @staticmethod
    def forward(ctx, x, cls_token, pos, x_hi=None, pack_only=False, gather=None):
        # x_hi: optional second half of the batch (abnormal sequences); the cat is fused into the kernel
        # pack_only: the bf16 activation stream - the result exists only as layer 0's packed operand (returned as its bf16 view)
        # gather = (clip_idx int64 [N * Lc] on the device, N, Lc): x is a feature BANK [clips, P, d] and sequence n the clips
        #   clip_idx[n*Lc : (n+1)*Lc] - batch formation, cat and CLS concat in one pass (lstc_cls_concat_gather_fwd)
        if gather is not None:
            return ClsConcatFunction._forward_gather(ctx, x, cls_token, pos, pack_only, gather)
        N_lo, Sm1, dm = x.shape
        N = N_lo + (x_hi.shape[0] if x_hi is not None else 0)
        S = Sm1 + 1
        x = x.contiguous()
        x_hi = x_hi.contiguous() if x_hi is not None else None
        ctx.pack_only = bool(pack_only)
        if pack_only:
            if cls_token is not None or pos is not None or ctx.needs_input_grad[0] or not act_rows_ok(N * S, dm):
                raise RuntimeError("ClsConcatFunction(pack_only): nothing upstream may need a gradient and [N*S, d] must fit the bf16 stream")
            lib = _lib.load()
            buf = torch.empty((int(lib.lstc_pack1_bytes(N * S, dm)),), device=x.device, dtype=torch.uint8)
            check(lib.lstc_cls_concat_fwd_pack(dev_ptr(x), dev_ptr(x_hi), N_lo, None, None, None, N, S, dm, dev_ptr(buf),
                                               stream_ptr()), "lstc_cls_concat_fwd_pack")
            ctx.two = x_hi is not None
            ctx.meta = (N, S, dm, False, None)
            return buf.view(torch.bfloat16)
        y = torch.empty((N, S, dm), device=x.device, dtype=torch.float32)
        pos_s = pos[0, :S].contiguous() if pos is not None else None
        cls_v = cls_token.reshape(-1) if cls_token is not None else None
        if _fused_pack_shape(N * S, dm):
            # bf16 mode: the kernel also writes layer 0's packed A operand
            lib = _lib.load()
            buf = torch.empty((int(lib.lstc_pack1_bytes(N * S, dm)),), device=x.device, dtype=torch.uint8)
            check(lib.lstc_cls_concat_fwd_pack(dev_ptr(x), dev_ptr(x_hi), N_lo, dev_ptr(cls_v), dev_ptr(pos_s), dev_ptr(y),
                                               N, S, dm, dev_ptr(buf), stream_ptr()), "lstc_cls_concat_fwd_pack")
            _register_pack(y.view(N * S, dm), Packed(buf, N * S, dm, _lib.BF16P))
        else:
            check(_lib.load().lstc_cls_concat_fwd(dev_ptr(x), dev_ptr(x_hi), N_lo, dev_ptr(cls_v), dev_ptr(pos_s), dev_ptr(y),
                                                  N, S, dm, stream_ptr()), "lstc_cls_concat_fwd")
        ctx.two = x_hi is not None
        ctx.meta = (N, S, dm, cls_token is not None, pos.shape if pos is not None else None)
        return y

    _forward_gather = None          # bound below: functional._cls_concat_gather

    @staticmethod
    def backward(ctx, dy):
        N, S, dm, learned, pos_shape = ctx.meta
        if ctx.pack_only:
            return None, None, None, None, None, None
        dy = dy.contiguous()
        dx = dcls = dpos = None
        if ctx.needs_input_grad[0]:
            if ctx.two:
                raise RuntimeError("input gradients with a split batch are not supported: concatenate first")
            dx = torch.empty((N, S - 1, dm), device=dy.device, dtype=torch.float32)
            check(_lib.load().lstc_cls_concat_bwd(dev_ptr(dy), dev_ptr(dx), N, S, dm, int(not learned), stream_ptr()),
                  "lstc_cls_concat_bwd")
        if learned or pos_shape is not None:
            tok = colsum(dy.view(N, S * dm))                 # sum over sequences of every (token, channel)
            if learned:
                dcls = tok[:dm].clone().view(1, 1, dm)
            if pos_shape is not None:
                dpos = torch.zeros(pos_shape, device=dy.device, dtype=torch.float32)
                dpos[0, :S] = tok.view(S, dm)
        return dx, dcls, dpos, None, None, None


def _cls_concat_gather(ctx, bank, cls_token, pos, pack_only, gather):
    idx, N, Lc = gather
    clips, P, dm = bank.shape
    S = Lc * P + 1
    if bank.dtype != torch.float32 or not bank.is_contiguous() or idx.dtype != torch.int64 or idx.numel() != N * Lc or bank.requires_grad:
        raise RuntimeError("ClsConcatFunction(gather): bank must be a contiguous float32 [clips, P, d] tensor without gradient and idx int64 [N * Lc]")
    lib = _lib.load()
    want_pack = pack_only or _fused_pack_shape(N * S, dm)
    if pack_only and (cls_token is not None or pos is not None or not act_rows_ok(N * S, dm)):
        raise RuntimeError("ClsConcatFunction(pack_only): nothing upstream may need a gradient and [N*S, d] must fit the bf16 stream")
    y = None if pack_only else torch.empty((N, S, dm), device=bank.device, dtype=torch.float32)
    buf = torch.empty((int(lib.lstc_pack1_bytes(N * S, dm)),), device=bank.device, dtype=torch.uint8) if want_pack else None
    pos_s = pos[0, :S].contiguous() if pos is not None else None
    cls_v = cls_token.reshape(-1) if cls_token is not None else None
    check(lib.lstc_cls_concat_gather_fwd(dev_ptr(bank), clips, dev_ptr(idx.contiguous()), P, dev_ptr(cls_v), dev_ptr(pos_s), dev_ptr(y),
                                         N, S, dm, dev_ptr(buf), stream_ptr()), "lstc_cls_concat_gather_fwd")
    ctx.pack_only = bool(pack_only)
    ctx.two = False
    ctx.meta = (N, S, dm, cls_token is not None, pos.shape if pos is not None else None)
    if pack_only:
        return buf.view(torch.bfloat16)
    if want_pack:
        _register_pack(y.view(N * S, dm), Packed(buf, N * S, dm, _lib.BF16P))
    return y


ClsConcatFunction._forward_gather = staticmethod(_cls_concat_gather)


class DropoutFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, site):
        seed = next_seed()
        _note(site, p, seed, x.shape)
        ctx.ps = (p, seed)
        return dropout_apply(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return dropout_apply(dy, *ctx.ps), None, None


class HeadFunction(torch.autograd.Function):
    """Regressor (models/Regressor.py:7-21) / Classifier (models/Classifier.py:8-23):
    view(-1,d) -> Linear(d,512)+ReLU+Drop -> Linear(512,32)+Drop -> Linear(32,c) -> Sigmoid | Softmax."""

    @staticmethod
    @_small_m_fwd
    def forward(ctx, x, w0, b0, w3, b3, w5, b5, cfg):
        x2 = x.contiguous().view(-1, x.shape[-1])
        p = cfg["dropout"] if cfg["training"] else 0.0
        c = w5.shape[0]
        s1 = next_seed() if p > 0 else 0
        s2 = next_seed() if p > 0 else 0
        rows = x2.shape[0]
        if p > 0:
            _note(cfg["site"] + ".2", p, s1, (rows, w0.shape[0]))
            _note(cfg["site"] + ".4", p, s2, (rows, 32))
        h1 = gemm(x2, w0, trans_b=True, bias=b0, relu=True, dropout=(p, s1))        # relu then dropout
        h2 = gemm(h1, w3, trans_b=True, bias=b3, dropout=(p, s2))
        out = torch.empty((rows, c), device=x.device, dtype=torch.float32)
        check(_lib.load().lstc_head_out_fwd(dev_ptr(h2), dev_ptr(w5), dev_ptr(b5), dev_ptr(out), rows, c, stream_ptr()),
              "lstc_head_out_fwd")
        ctx.cfg = dict(cfg, p=p, s1=s1, s2=s2, c=c, in_shape=tuple(x.shape))
        ctx.save_for_backward(x2, w0, w3, w5, h1, h2, out)
        return out

    @staticmethod
    @_small_m_bwd
    def backward(ctx, dout):
        x2, w0, w3, w5, h1, h2, out = ctx.saved_tensors
        cf = ctx.cfg
        rows, c, p = x2.shape[0], cf["c"], cf["p"]
        dout = dout.contiguous()
        dh2 = torch.empty_like(h2)
        dw5 = torch.empty_like(w5)                 # written by the kernel (one workgroup, fixed order): no fill
        db5 = torch.empty((c,), device=x2.device, dtype=torch.float32)
        check(_lib.load().lstc_head_out_bwd(dev_ptr(h2), dev_ptr(w5), dev_ptr(out), dev_ptr(dout), dev_ptr(dh2),
                                            dev_ptr(dw5), dev_ptr(db5), rows, c, stream_ptr()), "lstc_head_out_bwd")
        da2 = dropout_apply(dh2, p, cf["s2"]) if p > 0 else dh2          # grad of Linear(512,32) output
        db3 = colsum(da2)
        dw3 = wgrad(da2, h1)
        # h1 = dropout(relu(.)) was saved: (h1 > 0) <=> relu passed AND the unit was kept, and dropped units get a
        # zero gradient from dropout_apply anyway, so masking with h1 > 0 (fused in the GEMM epilogue) is exact.
        da1 = gemm(da2, w3, relu_mask=h1)                                # [rows, 512]
        if p > 0:
            da1 = dropout_apply(da1, p, cf["s1"])
        db0 = colsum(da1)
        dw0 = deliver(w0, wgrad(da1, x2, out=grad_sink(w0)))
        dx = gemm(da1, w0).view(cf["in_shape"]) if ctx.needs_input_grad[0] else None
        return dx, dw0, db0, dw3, db3, dw5, db5, None


class VadLossFunction(torch.autograd.Function):
    """MIL (+CE | +BCE) loss: one lstc_vad_loss launch produces the scalars and d(loss)/d(head output)."""

    @staticmethod
    def forward(ctx, out, abn_labels, targets, cfg):
        out = out.contiguous()
        dev = out.device
        d = LossDesc()
        d.mode = cfg["mode"]
        d.bs_global, d.bs_local, d.rank_off = cfg["bs_global"], cfg["bs_local"], cfg["rank_off"]
        d.part_num, d.score_len, d.label_len, d.l1_skip = cfg["part_num"], cfg["score_len"], cfg["label_len"], cfg["l1_skip"]
        d.lambda_1, d.lambda_MIL, d.lambda_aux = cfg["lambda_1"], cfg["lambda_MIL"], cfg["lambda_aux"]
        d.lambda_normal, d.lambda_abnormal = cfg.get("lambda_normal", 0.0), cfg.get("lambda_abnormal", 0.0)
        dout = torch.empty_like(out)
        scalars = torch.empty((5,), device=dev, dtype=torch.float32)
        # the bag vector is an exchange buffer: only a rank-sharded batch needs it (zero-filled: the other ranks' slots are summed in)
        sharded = cfg["bs_local"] != cfg["bs_global"]
        bag = torch.zeros((2 * cfg["bs_global"],), device=dev, dtype=torch.float32) if sharded else None
        labs = abn_labels.contiguous().float() if abn_labels is not None else None
        tg = targets.contiguous().float() if targets is not None else None
        d.out, d.abn_labels, d.targets, d.bag = dev_ptr(out), dev_ptr(labs), dev_ptr(tg), dev_ptr(bag)
        d.dout, d.scalars = dev_ptr(dout), dev_ptr(scalars)
        lib = _lib.load()
        group = cfg.get("group")
        if cfg["bs_local"] == cfg["bs_global"]:
            d.phase = 2
            check(lib.lstc_vad_loss(C.byref(d), stream_ptr()), "lstc_vad_loss")
        else:
            # rank-sharded batch: phase 0 writes this rank's bag maxima into its slots of the zero-padded global vector,
            # the ranks SUM it (2*bs floats: the only coupling between ranks besides the gradient all-reduce), phase 1
            # evaluates the hinge of the local videos against the global vector.  ``cfg["exchange"]`` replaces the
            # collective in tests that emulate several ranks on one device.
            exchange = cfg.get("exchange")
            d.phase = 0
            check(lib.lstc_vad_loss(C.byref(d), stream_ptr()), "lstc_vad_loss")
            if exchange is not None:
                exchange(bag)
            else:
                from .dist import all_reduce_sum
                all_reduce_sum(bag, group)
            d.phase = 1
            check(lib.lstc_vad_loss(C.byref(d), stream_ptr()), "lstc_vad_loss")
        ctx.save_for_backward(dout)
        ctx.mark_non_differentiable(scalars)
        return scalars[0], scalars

    @staticmethod
    def backward(ctx, gloss, _gs):
        (dout,) = ctx.saved_tensors
        return dout * gloss, None, None, None


# Activation packs (f32x3 mode) are shared between the GEMMs of one forward / backward body.
def _with_pack_memo(fn):
    def wrapped(*args, **kwargs):
        with pack_memo():
            return fn(*args, **kwargs)
    wrapped.__doc__ = fn.__doc__
    return wrapped


for _cls in (MHAFunction, MHAClsFunction, MHAClsAssocFunction, FFNFunction, HeadFunction):
    _cls.forward = staticmethod(_with_pack_memo(_cls.forward))
    _cls.backward = staticmethod(_with_pack_memo(_cls.backward))
