"""Adagrad with the reference's settings, stepped by the fused HIP kernel.

The reference builds ``torch.optim.Adagrad([{params: encoder, lr: lr_encoder}, {params: head, lr: lr_head}],
weight_decay=wd)`` (Train/temporal_transformer_shanghaitech.py:83-85) and calls ``zero_grad / backward /
[clip_grad_norm_] / step`` (:137-142).  This class keeps that surface (param groups, ``zero_grad``, ``step``,
``state_dict``) and the exact update ``g += wd*w; s += g*g; w -= lr*g/(sqrt(s)+1e-10)``; parameters whose
``grad`` is None are skipped exactly like upstream (unused LayerNorms never get a gradient).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, dev_ptr, stream_ptr


class Adagrad(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, lr_decay=0, weight_decay=0, initial_accumulator_value=0, eps=1e-10):
        if lr_decay != 0:
            raise NotImplementedError("lr_decay is 0 everywhere on the LSTC_VAD path")
        defaults = dict(lr=lr, weight_decay=weight_decay, eps=eps, initial_accumulator_value=initial_accumulator_value)
        super().__init__(params, defaults)
        for group in self.param_groups:
            for p in group["params"]:
                self.state[p]["sum"] = torch.full_like(p, float(group["initial_accumulator_value"]),
                                                       memory_format=torch.preserve_format)
                self.state[p]["step"] = 0

    @torch.no_grad()
    def step(self, closure=None, grad_scales=None, only=None):
        """``grad_scales``: optional {group index: device-or-host scale} from ``clip_grad_norm_`` below.  ``only``: step just these
        parameters (one gradient bucket of a data-parallel job whose all-reduce has landed - engine.TrainStep steps bucket k while
        the backward of the layers below it is still running); the element arithmetic does not depend on how the parameters are
        grouped into launches, so any partition into ``only`` sets gives the weights of ONE whole step bit for bit."""
        lib = _lib.load()
        items, keep, updated = [], [], []
        sel = None if only is None else {id(p) for p in only}
        for gi, group in enumerate(self.param_groups):
            gs = 1.0 if not grad_scales else float(grad_scales.get(gi, 1.0))
            for p in group["params"]:
                if p.grad is None or (sel is not None and id(p) not in sel):
                    continue
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                state = self.state[p]
                state["step"] += 1
                keep.append(g)
                updated.append(p)
                items.append((dev_ptr(p.data), dev_ptr(g), dev_ptr(state["sum"]), p.numel(), float(group["lr"]),
                              float(group["weight_decay"]), float(group["eps"]), gs))
        if items:           # every parameter in ONE launch (lstc_adagrad_multi; element arithmetic of lstc_adagrad_step)
            arr = (_lib.AdagradItem * len(items))(*items)
            check(lib.lstc_adagrad_multi(arr, len(items), stream_ptr()), "lstc_adagrad_multi")
        from .functional import bump_weight_epoch, repack_weights
        bump_weight_epoch(updated)   # THESE weights changed through raw pointers: their packed copies (f32x3 / bf16 GEMM) are stale
        if items and not torch.cuda.is_current_stream_capturing():
            # bf16 mode: every packed weight copy of the finished step rebuilt in one launch (a captured step keeps the
            # lazily issued packs of its forward instead: the graph replays those)
            repack_weights(updated)
        return None


def clip_grad_norm_(parameters, max_norm: float, norm_type: float = 2.0) -> torch.Tensor:
    """``torch.nn.utils.clip_grad_norm_(params, 10)`` (Train/temporal_transformer_shanghaitech.py:139-141): total L2 norm over
    the given parameters' grads and an in-place scale by ``max_norm / (total + 1e-6)`` clamped to 1 (torch semantics) - as TWO
    launches over the whole list (``lstc_sqnorm_multi`` + ``lstc_clip_scale_multi``, the coefficient formed on the device)
    instead of upstream's kernel per tensor and host comparison.  No host sync, so a captured step may contain it.  Returns
    the total norm as a 0-dim device tensor (torch returns a device tensor too)."""
    if norm_type != 2.0:
        raise NotImplementedError("clip_grad_norm_: only the L2 norm the LSTC_VAD scripts use")
    params = [p for p in parameters if p.grad is not None and p.grad.numel() > 0]       # torch accepts empty tensors: nothing to scale
    if not params:
        return torch.zeros(())
    for p in params:
        if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
            raise RuntimeError("clip_grad_norm_: gradients must be contiguous float32 tensors")
    lib = _lib.load()
    dev = params[0].grad.device
    items = (_lib.VecItem * len(params))(*[(dev_ptr(p.grad), p.grad.numel()) for p in params])
    need = int(lib.lstc_sqnorm_multi_scratch(items, len(params)))
    scratch = torch.empty((need + 2,), device=dev, dtype=torch.float32)       # [partials..., sum of squares, total norm]
    sq = scratch[need:]
    check(lib.lstc_sqnorm_multi(items, len(params), dev_ptr(scratch), need, dev_ptr(sq), stream_ptr()), "lstc_sqnorm_multi")
    check(lib.lstc_clip_scale_multi(items, len(params), dev_ptr(sq), float(max_norm), stream_ptr()), "lstc_clip_scale_multi")
    return sq[1]
