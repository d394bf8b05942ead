"""Loss functions of the LSTC_VAD train loops on top of the single fused HIP loss kernel.

``training_loss`` is what the Train/*.py scripts and bench.py call: ONE launch gives the five scalars the
reference logs and d(loss)/d(head output).  The reference-named functions (``get_MIL_loss``,
``get_CE_loss``, ``get_BCE_loss``) keep their signatures for drop-in callers; each is the
same kernel with the other terms switched off.
"""
from __future__ import annotations

import torch

from .functional import VadLossFunction

MODE_STN, MODE_LTN, MODE_STN_BCE = 0, 1, 2


def _dist_info(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def training_loss(args, mode: str, head_out: torch.Tensor, abnorm_labs=None, group=None, distributed=None, exchange=None):
    """Loss of one training step from the head output of THIS rank's sequences.

    mode "LTN": Train/temporal_transformer_shanghaitech.py:103-134 (MIL on out[:,1] + lambda_CE * CE unless
                args.temporal_only); "STN": Train/spatio_transformer_shanghaitech.py:99-101;
                "STN_MIL_CE": Train/spatio_transformer_MIL_CE.py:176-181.
    ``args.batch_size`` is this rank's number of normal/abnormal pairs; under data parallelism the global
    batch is ``world_size * args.batch_size`` pairs and the hinge couples all of them.
    ``distributed=(rank, world)`` overrides the process-group lookup and ``exchange(bag)`` the bag all-reduce (tests that
    emulate ranks on one device).  Returns ``(loss, scalars)`` with scalars = [loss, MIL, err, l1, aux] (rank-local
    contributions: summing them over ranks gives the single-process values)."""
    rank, world = _dist_info(group) if distributed is None else distributed
    bs_l, pn, L = args.batch_size, args.part_num, args.part_len
    bs_g = bs_l * world
    cfg = dict(bs_global=bs_g, bs_local=bs_l, rank_off=rank * bs_l, part_num=pn, label_len=L,
               lambda_1=float(args.lambda_1), lambda_normal=0.0, lambda_abnormal=0.0, group=group, exchange=exchange)
    if mode == "LTN":
        use_ce = not getattr(args, "temporal_only", False)
        cfg.update(mode=MODE_LTN, score_len=1, l1_skip=bs_g, lambda_MIL=float(args.lambda_MIL),
                   lambda_aux=float(args.lambda_CE) if use_ce else 0.0)
        labs = abnorm_labs if use_ce else None
    elif mode == "STN":
        cfg.update(mode=MODE_STN, score_len=L, l1_skip=bs_g * pn * L, lambda_MIL=1.0, lambda_aux=0.0)
        labs = None
    elif mode == "STN_MIL_CE":
        cfg.update(mode=MODE_STN_BCE, score_len=L, l1_skip=bs_g, lambda_MIL=1.0, lambda_aux=float(args.lambda_BCE),
                   lambda_normal=float(args.lambda_normal), lambda_abnormal=float(args.lambda_abnormal))
        labs = abnorm_labs
    else:
        raise ValueError(mode)
    out2 = head_out.reshape(-1, 2 if mode == "LTN" else 1)
    return VadLossFunction.apply(out2, labs, None, cfg)


# ------------------------------------------------------------------ reference-named entry points (single rank)
def get_MIL_loss(args, y_pred, part_len=None):
    """(loss, err, l1).  Covers the three reference variants: STN ``get_MIL_loss(args, y[2bs, pn*L, 1])``
    (Train/spatio_transformer_shanghaitech.py:21-32), LTN ``get_MIL_loss(args, y[2bs*pn])``
    (Train/temporal_transformer_shanghaitech.py:25-36) and co-teaching ``get_MIL_loss(args, y, part_len)``
    (Train/spatio_transformer_MIL_CE.py:32-44).  ``y_pred[batch_size:]`` slices the first dim, as upstream."""
    bs, pn = args.batch_size, args.part_num
    per_video = y_pred.numel() // (2 * bs)
    L = part_len if part_len is not None else per_video // pn
    skip = bs * (y_pred.numel() // y_pred.shape[0])
    cfg = dict(mode=MODE_STN, bs_global=bs, bs_local=bs, rank_off=0, part_num=pn, score_len=L, label_len=L,
               l1_skip=skip, lambda_1=float(args.lambda_1), lambda_MIL=1.0, lambda_aux=0.0)
    loss, sc = VadLossFunction.apply(y_pred.reshape(-1, 1), None, None, cfg)
    return loss, sc[2], sc[3]


def get_CE_loss(args, outputs, labs):
    """``F.cross_entropy(outputs, labs)`` on softmax *outputs* with soft targets
    (Train/temporal_transformer_shanghaitech.py:21-23)."""
    rows = outputs.shape[0]
    bs = rows // (2 * args.part_num)
    cfg = dict(mode=MODE_LTN, bs_global=bs, bs_local=bs, rank_off=0, part_num=args.part_num, score_len=1,
               label_len=1, l1_skip=0, lambda_1=0.0, lambda_MIL=0.0, lambda_aux=1.0)
    loss, _ = VadLossFunction.apply(outputs, None, labs.reshape(rows, 2), cfg)
    return loss


def get_BCE_loss(args, outputs, labs):
    """Weighted BCE on part-level scores ``outputs`` [2bs, pn] against ``labs`` [2bs, pn, 2]
    (Train/spatio_transformer_MIL_CE.py:23-26)."""
    bs2, pn = outputs.shape
    cfg = dict(mode=MODE_STN_BCE, bs_global=bs2 // 2, bs_local=bs2 // 2, rank_off=0, part_num=pn, score_len=1,
               label_len=1, l1_skip=0, lambda_1=0.0, lambda_MIL=0.0, lambda_aux=1.0,
               lambda_normal=float(args.lambda_normal), lambda_abnormal=float(args.lambda_abnormal))
    loss, _ = VadLossFunction.apply(outputs.reshape(-1, 1), None, labs.reshape(-1, 2), cfg)
    return loss
