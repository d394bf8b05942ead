"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (torch-CPU float32, functional style) of the LSTC_VAD training hot
path that ``BASELINE.json.north_star`` names.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / reported CPU baseline.  The product (``lstc_vad_amd``)
never routes through this file and raises when its HIP library is missing.

Parity status: PINNED.  The reference ships no tests or golden vectors of its own
(SURVEY.md 4, 8c), so this restatement is pinned against outputs of the reference
itself: ``tests/golden/make_golden.py`` imports the unmodified reference classes
from /root/reference on CPU (with ``cv2`` / ``h5py`` stubbed, both unused by the
path), runs them on the inputs of ``lstc_vad_amd.synthetic`` and commits the
results under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
file against those fixtures.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Parameters travel in a plain ``dict`` whose keys are the
reference ``state_dict`` keys, so a reference ``state_dict()`` can be
fed in directly.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class EncoderCfg:
    """Constructor arguments of the reference ``Encoder`` (models/Encoder.py:6-11)."""
    n_layers: int = 3
    n_head: int = 8
    d_k: int = 256
    d_v: int = 256
    d_model: int = 2048
    d_inner: int = 4096
    MHA_attn_dropout: float = 0.1
    MHA_fc_dropout: float = 0.1
    MHA_layerNorm: bool = False
    FFN_dropout: float = 0.1
    FFN_layerNorm: bool = True
    CLS_learned: bool = False
    position_dropout: float = 0.1
    position_encoding: bool = False
    max_position_tokens: int = 100
    relative_pe: bool = False
    window_size: int = 4
    window_depth: int = 3
    input_layerNorm: bool = False
    relative_pe_2D: bool = False
    FFN_need: bool = True


# ------------------------------------------------------------ relative position index
def relative_position_index_3d(window_depth: int, window_size: int) -> Tensor:
    """Closed form of the buffer built at models/MultiHeadAttention.py:56-73.

    token t -> (d, h, w) = (t // ws^2, (t // ws) % ws, t % ws);
    index[i, j] = (d_i-d_j+L-1)(2ws-1)^2 + (h_i-h_j+ws-1)(2ws-1) + (w_i-w_j+ws-1).
    """
    L, ws = window_depth, window_size
    t = torch.arange(L * ws * ws)
    d, h, w = t // (ws * ws), (t // ws) % ws, t % ws
    q = 2 * ws - 1
    return ((d[:, None] - d[None, :] + L - 1) * q * q
            + (h[:, None] - h[None, :] + ws - 1) * q
            + (w[:, None] - w[None, :] + ws - 1)).to(torch.int64)


def relative_position_index_2d(window_size: int) -> Tensor:
    """Closed form of the 2-D variant, models/MultiHeadAttention.py:76-89."""
    ws = window_size
    t = torch.arange(ws * ws)
    h, w = t // ws, t % ws
    q = 2 * ws - 1
    return ((h[:, None] - h[None, :] + ws - 1) * q + (w[:, None] - w[None, :] + ws - 1)).to(torch.int64)


# ----------------------------------------------------------------------- param shapes
def encoder_param_shapes(cfg: EncoderCfg) -> Dict[str, tuple]:
    """Parameter names/shapes of the reference Encoder in ``named_parameters()`` order
    (models/Encoder.py:15-31, models/MultiHeadAttention.py:40-47,55-57,77-78, models/FFN.py:8-10)."""
    shapes: Dict[str, tuple] = {}
    if cfg.CLS_learned:
        shapes["cls_token"] = (1, 1, cfg.d_model)
    if cfg.position_encoding:
        shapes["position_enc"] = (1, cfg.max_position_tokens, cfg.d_model)
    for i in range(cfg.n_layers):
        p = f"layer_stack.{i}."
        if cfg.relative_pe_2D:
            # models/MultiHeadAttention.py:76-78 (2-D overrides 3-D when both are set)
            shapes[p + "slf_attn.relative_position_bias_table"] = ((2 * cfg.window_size - 1) ** 2, cfg.n_head)
        elif cfg.relative_pe:
            shapes[p + "slf_attn.relative_position_bias_table"] = (
                (2 * cfg.window_depth - 1) * (2 * cfg.window_size - 1) ** 2, cfg.n_head)
        shapes[p + "slf_attn.w_qs.weight"] = (cfg.n_head * cfg.d_k, cfg.d_model)
        shapes[p + "slf_attn.w_ks.weight"] = (cfg.n_head * cfg.d_k, cfg.d_model)
        shapes[p + "slf_attn.w_vs.weight"] = (cfg.n_head * cfg.d_v, cfg.d_model)
        shapes[p + "slf_attn.fc.weight"] = (cfg.d_model, cfg.n_head * cfg.d_v)
        shapes[p + "slf_attn.layer_norm.weight"] = (cfg.d_model,)
        shapes[p + "slf_attn.layer_norm.bias"] = (cfg.d_model,)
        shapes[p + "pos_ffn.w_1.weight"] = (cfg.d_inner, cfg.d_model)
        shapes[p + "pos_ffn.w_1.bias"] = (cfg.d_inner,)
        shapes[p + "pos_ffn.w_2.weight"] = (cfg.d_model, cfg.d_inner)
        shapes[p + "pos_ffn.w_2.bias"] = (cfg.d_model,)
        shapes[p + "pos_ffn.layer_norm.weight"] = (cfg.d_model,)
        shapes[p + "pos_ffn.layer_norm.bias"] = (cfg.d_model,)
    shapes["layer_norm.weight"] = (cfg.d_model,)
    shapes["layer_norm.bias"] = (cfg.d_model,)
    return shapes


def head_param_shapes(d_model: int, kind: str, hidden_dim: int = 512) -> Dict[str, tuple]:
    """``Regressor`` (models/Regressor.py:7-9) / ``Classifier`` (models/Classifier.py:8-10)."""
    pre = "regressor" if kind == "regressor" else "classifier"
    c = 1 if kind == "regressor" else 2
    return {f"{pre}.0.weight": (hidden_dim, d_model), f"{pre}.0.bias": (hidden_dim,),
            f"{pre}.3.weight": (32, hidden_dim), f"{pre}.3.bias": (32,),
            f"{pre}.5.weight": (c, 32), f"{pre}.5.bias": (c,)}


# --------------------------------------------------------------------------- dropout
def _drop(x: Tensor, p: float, training: bool, masks: Optional[dict], name: str) -> Tensor:
    """Dropout site.  ``masks[name]`` (a 0/1 keep tensor) is injected when given, so a
    HIP run with its own counter-based RNG can be replayed exactly; otherwise torch's
    generator is used like the reference's ``nn.Dropout``."""
    if not training or p <= 0.0:
        return x
    if masks is not None and name in masks:
        return x * (masks[name].to(x.dtype) * (1.0 / (1.0 - p)))
    return F.dropout(x, p, True)


# ---------------------------------------------------------------------- encoder pieces
def mha_forward(P: Dict[str, Tensor], pre: str, x: Tensor, cfg: EncoderCfg, training: bool,
                masks: Optional[dict] = None, return_attn: bool = False):
    """models/MultiHeadAttention.py:93-132 (self-attention: q=k=v=x, mask never passed)."""
    N, S, _ = x.shape
    H, dk, dv = cfg.n_head, cfg.d_k, cfg.d_v
    q = (x @ P[pre + "w_qs.weight"].t()).view(N, S, H, dk).transpose(1, 2)     # :97,:101
    k = (x @ P[pre + "w_ks.weight"].t()).view(N, S, H, dk).transpose(1, 2)     # :98
    v = (x @ P[pre + "w_vs.weight"].t()).view(N, S, H, dv).transpose(1, 2)     # :99
    logits = (q / (dk ** 0.5)) @ k.transpose(2, 3)                             # :103 scale Q first
    if cfg.relative_pe:                                                        # :106-111
        idx = P[pre + "relative_position_index"][: S - 1, : S - 1].reshape(-1)
        bias = P[pre + "relative_position_bias_table"][idx].reshape(S - 1, S - 1, H).permute(2, 0, 1)
        logits = torch.cat([logits[:, :, :1, :],
                            torch.cat([logits[:, :, 1:, :1], logits[:, :, 1:, 1:] + bias.unsqueeze(0)], 3)], 2)
    if cfg.relative_pe_2D:                                                     # :113-117 (needs S-1 == ws*ws)
        idx = P[pre + "relative_position_index"].reshape(-1)
        ww = cfg.window_size * cfg.window_size
        bias = P[pre + "relative_position_bias_table"][idx].reshape(ww, ww, H).permute(2, 0, 1)
        logits = torch.cat([logits[:, :, :1, :],
                            torch.cat([logits[:, :, 1:, :1], logits[:, :, 1:, 1:] + bias.unsqueeze(0)], 3)], 2)
    attn = torch.softmax(logits, dim=-1)                                       # :119
    attn_d = _drop(attn, cfg.MHA_attn_dropout, training, masks, pre + "attn_dropout")
    o = (attn_d @ v).transpose(1, 2).reshape(N, S, H * dv)                     # :120-122
    y = _drop(o @ P[pre + "fc.weight"].t(), cfg.MHA_fc_dropout, training, masks, pre + "dropout")  # :123
    y = y + x                                                                  # :124
    if cfg.MHA_layerNorm:                                                      # :125-126 eps=1e-6 (:47)
        y = F.layer_norm(y, (cfg.d_model,), P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], 1e-6)
    return (y, attn_d) if return_attn else (y, None)


def ffn_forward(P: Dict[str, Tensor], pre: str, x: Tensor, cfg: EncoderCfg, training: bool,
                masks: Optional[dict] = None) -> Tensor:
    """models/FFN.py:14-22."""
    h = torch.relu(x @ P[pre + "w_1.weight"].t() + P[pre + "w_1.bias"])
    y = h @ P[pre + "w_2.weight"].t() + P[pre + "w_2.bias"]
    y = _drop(y, cfg.FFN_dropout, training, masks, pre + "dropout") + x
    if cfg.FFN_layerNorm:
        y = F.layer_norm(y, (cfg.d_model,), P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], 1e-6)
    return y


def encoder_forward(P: Dict[str, Tensor], x: Tensor, cfg: EncoderCfg, training: bool = False,
                    masks: Optional[dict] = None, return_attn: bool = False):
    """models/Encoder.py:43-74.  ``x`` is ``[N, S-1, d_model]``; returns ``[N, S, d_model]``."""
    if cfg.input_layerNorm:                                                    # :48-49
        x = F.layer_norm(x, (cfg.d_model,), P["layer_norm.weight"], P["layer_norm.bias"], 1e-6)
    if cfg.CLS_learned:                                                        # :51-52
        cls = P["cls_token"].expand(x.shape[0], -1, -1)
    else:                                                                      # :54
        cls = x.mean(dim=1, keepdim=True)
    x = torch.cat([cls, x], dim=1)                                             # :55
    if cfg.position_encoding:                                                  # :57-59
        x = x + P["position_enc"][:, : x.shape[1], :]
        x = _drop(x, cfg.position_dropout, training, masks, "position_dropout")
    attns = []
    for i in range(cfg.n_layers):                                              # :61-67, EncoderLayer.py:18-30
        x, a = mha_forward(P, f"layer_stack.{i}.slf_attn.", x, cfg, training, masks, return_attn)
        if cfg.FFN_need:
            x = ffn_forward(P, f"layer_stack.{i}.pos_ffn.", x, cfg, training, masks)
        if return_attn:
            attns.append(a)
    return (x, attns) if return_attn else x


def head_forward(P: Dict[str, Tensor], feats: Tensor, kind: str, dropout_rate: float = 0.6,
                 training: bool = False, masks: Optional[dict] = None) -> Tensor:
    """``Regressor.forward`` (models/Regressor.py:18-21) or ``Classifier.forward``
    (models/Classifier.py:20-23): view(-1,d) -> Linear-ReLU-Drop -> Linear-Drop -> Linear -> Sigmoid|Softmax."""
    pre = "regressor" if kind == "regressor" else "classifier"
    x = feats.reshape(-1, feats.shape[-1])
    x = torch.relu(x @ P[f"{pre}.0.weight"].t() + P[f"{pre}.0.bias"])
    x = _drop(x, dropout_rate, training, masks, f"{pre}.2")
    x = x @ P[f"{pre}.3.weight"].t() + P[f"{pre}.3.bias"]
    x = _drop(x, dropout_rate, training, masks, f"{pre}.4")
    x = x @ P[f"{pre}.5.weight"].t() + P[f"{pre}.5.bias"]
    return torch.sigmoid(x) if kind == "regressor" else torch.softmax(x, dim=-1)


# ----------------------------------------------------------------------------- losses
def mil_loss(y_pred: Tensor, batch_size: int, part_num: int, part_len: int, lambda_1: float):
    """MIL ranking loss.  STN form: Train/spatio_transformer_shanghaitech.py:21-32 (mean over
    part_len, max over part_num); LTN form: Train/temporal_transformer_shanghaitech.py:25-36
    is the same with part_len=1.  ``y_pred[batch_size:]`` slices the *first* dim of whatever
    shape the caller passes (flat-slice quirk of the LTN / co-teach callers)."""
    bag = y_pred.reshape(batch_size * 2, part_num, part_len).mean(dim=-1).max(dim=-1)[0]
    nor, abn = bag[:batch_size], bag[batch_size:]
    err = torch.relu(1.0 - abn[None, :] + nor[:, None]).sum() / float(batch_size) ** 2
    l1 = y_pred[batch_size:].mean()
    return err + lambda_1 * l1, err, l1


def ce_loss(outputs: Tensor, labs: Tensor) -> Tensor:
    """Train/temporal_transformer_shanghaitech.py:21-23: ``F.cross_entropy`` applied to the
    classifier's *softmax outputs* with soft targets == mean_n(-sum_c t*log_softmax(p))."""
    return -(labs * torch.log_softmax(outputs, dim=-1)).sum(dim=-1).mean()


def bce_loss(outputs: Tensor, labs: Tensor, lambda_normal: float, lambda_abnormal: float) -> Tensor:
    """Train/spatio_transformer_MIL_CE.py:23-26."""
    return torch.mean(-lambda_normal * labs[:, :, 0] * torch.log(1 - outputs + 1e-8)
                      - lambda_abnormal * labs[:, :, 1] * torch.log(outputs + 1e-8))


def soft_targets(abnorm_labs: Tensor, batch_size: int, part_num: int, part_len: int) -> Tensor:
    """Train/temporal_transformer_shanghaitech.py:103-112: normal -> [1,0]; abnormal ->
    t1 = mean over part_len of the pseudo labels, t0 = 1 - t1; cat(normal, abnormal) -> [2bs, pn, 2]."""
    norm = torch.zeros(batch_size, part_num, 2)
    norm[:, :, 0] = 1.0
    t1 = abnorm_labs.reshape(batch_size, part_num, part_len).float().mean(dim=-1)
    abn = torch.stack([1.0 - t1, t1], dim=-1)
    return torch.cat([norm, abn], dim=0)


# ------------------------------------------------------------------------- optimizer
def adagrad_step(params, grads, states, lr: float, weight_decay: float, eps: float = 1e-10):
    """torch.optim.Adagrad defaults as used at Train/temporal_transformer_shanghaitech.py:83-85
    (lr_decay=0, initial_accumulator_value=0): g += wd*w; s += g*g; w -= lr*g/(sqrt(s)+eps).
    Entries whose grad is None are skipped, like the reference optimizer does."""
    for k, w in params.items():
        g = grads.get(k)
        if g is None:
            continue
        g = g + weight_decay * w
        states[k] = states[k] + g * g
        params[k] = w - lr * g / (states[k].sqrt() + eps)
    return params, states


def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float = 10.0) -> float:
    """torch.nn.utils.clip_grad_norm_ (Train/temporal_transformer_shanghaitech.py:139-141)."""
    gs = [g for g in grads.values() if g is not None]
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in gs)).item()
    coef = min(1.0, max_norm / (total + 1e-6))
    for k, g in grads.items():
        if g is not None:
            grads[k] = g * coef
    return total


# ------------------------------------------------------------------------ whole steps
@dataclass
class StepCfg:
    """The ``args`` fields the reference train loops read."""
    mode: str = "LTN"            # "LTN" | "STN" | "STN_MIL_CE"
    batch_size: int = 4          # normal/abnormal *pairs*
    part_num: int = 4
    part_len: int = 3
    n_patch: int = 16
    lambda_1: float = 0.01
    lambda_MIL: float = 1.0
    lambda_CE: float = 0.8
    lambda_BCE: float = 1.0
    lambda_normal: float = 0.2
    lambda_abnormal: float = 2.0
    temporal_only: bool = False
    head_dropout: float = 0.6
    lr_encoder: float = 1e-4
    lr_head: float = 1e-2
    weight_decay: float = 1e-3
    clip_grad: bool = False


def forward_loss(enc_P, head_P, enc_cfg: EncoderCfg, st: StepCfg, norm_feats, abnorm_feats, abnorm_labs,
                 training: bool = True, masks: Optional[dict] = None):
    """Forward + loss of one training step.

    LTN: Train/temporal_transformer_shanghaitech.py:103-134.
    STN: Train/spatio_transformer_shanghaitech.py:90-101.
    STN_MIL_CE: Train/spatio_transformer_MIL_CE.py:156-181 (round_i even branch, non-UCF)."""
    bs, pn, L, Pn, d = st.batch_size, st.part_num, st.part_len, st.n_patch, enc_cfg.d_model
    out = {}
    if st.mode == "LTN":
        nf = norm_feats.float().reshape(bs * pn, L * Pn, d)
        af = abnorm_feats.float().reshape(bs * pn, L * Pn, d)
        enc = encoder_forward(enc_P, torch.cat([nf, af], 0), enc_cfg, training, masks)
        cls = enc[:, 0, :].reshape(bs * 2, pn, d)
        outputs = head_forward(head_P, cls, "classifier", st.head_dropout, training, masks).reshape(bs * 2 * pn, -1)
        score = outputs[:, 1]
        if not st.temporal_only:
            labs = soft_targets(abnorm_labs, bs, pn, L).reshape(bs * 2 * pn, -1)
            ce = ce_loss(outputs, labs)
        else:
            ce = torch.zeros(())
        mil, err, l1 = mil_loss(score, bs, pn, 1, st.lambda_1)
        loss = st.lambda_MIL * mil + st.lambda_CE * ce
        out.update(outputs=outputs, score=score, mil=mil, err=err, l1=l1, aux=ce, loss=loss, enc_cls=enc[:, 0, :])
    elif st.mode == "STN":
        nf = norm_feats.float().reshape(bs * pn * L, Pn, d)
        af = abnorm_feats.float().reshape(bs * pn * L, Pn, d)
        enc = encoder_forward(enc_P, torch.cat([nf, af], 0), enc_cfg, training, masks)
        cls = enc[:, 0, :].reshape(bs * 2, pn * L, d)
        outputs = head_forward(head_P, cls, "regressor", st.head_dropout, training, masks).reshape(bs * 2, pn * L, -1)
        loss, err, l1 = mil_loss(outputs, bs, pn, L, st.lambda_1)
        out.update(outputs=outputs, score=outputs.reshape(-1), mil=loss, err=err, l1=l1,
                   aux=torch.zeros(()), loss=loss, enc_cls=enc[:, 0, :])
    elif st.mode == "STN_MIL_CE":
        labs = soft_targets(abnorm_labs, bs, pn, L)
        nf = norm_feats.float().reshape(bs * pn * L, Pn, d)
        af = abnorm_feats.float().reshape(bs * pn * L, Pn, d)
        enc = encoder_forward(enc_P, torch.cat([nf, af], 0), enc_cfg, training, masks)
        outputs = head_forward(head_P, enc[:, 0, :], "regressor", st.head_dropout, training, masks)   # [2bs*pn*L, 1]
        mil, err, l1 = mil_loss(outputs, bs, pn, L, st.lambda_1)      # flat-slice l1 quirk (:41-42)
        bce = bce_loss(outputs.reshape(bs * 2, pn, L).mean(dim=-1), labs, st.lambda_normal, st.lambda_abnormal)
        loss = st.lambda_BCE * bce + mil
        out.update(outputs=outputs, score=outputs.reshape(-1), mil=mil, err=err, l1=l1, aux=bce, loss=loss,
                   enc_cls=enc[:, 0, :])
    else:
        raise ValueError(st.mode)
    return out


def train_step(enc_P, head_P, enc_S, head_S, enc_cfg: EncoderCfg, st: StepCfg,
               norm_feats, abnorm_feats, abnorm_labs, masks: Optional[dict] = None, training: bool = True):
    """One full optimisation step (forward, loss, backward, [clip], Adagrad) on CPU.
    Follows Train/temporal_transformer_shanghaitech.py:103-142.  Parameters are given as
    dicts of plain tensors and returned updated; ``*_S`` are the Adagrad accumulators."""
    enc_leaf = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v)
                for k, v in enc_P.items()}
    head_leaf = {k: v.detach().clone().requires_grad_(True) for k, v in head_P.items()}
    out = forward_loss(enc_leaf, head_leaf, enc_cfg, st, norm_feats, abnorm_feats, abnorm_labs, training, masks)
    out["loss"].backward()
    enc_g = {k: v.grad for k, v in enc_leaf.items() if v.is_floating_point()}
    head_g = {k: v.grad for k, v in head_leaf.items()}
    if st.clip_grad:
        clip_grad_norm(enc_g, 10.0)
        clip_grad_norm(head_g, 10.0)
    with torch.no_grad():
        enc_new = {k: v.detach() for k, v in enc_leaf.items()}
        head_new = {k: v.detach() for k, v in head_leaf.items()}
        fl = {k: v for k, v in enc_new.items() if v.is_floating_point()}
        fl, enc_S = adagrad_step(fl, enc_g, enc_S, st.lr_encoder, st.weight_decay)
        enc_new.update(fl)
        head_new, head_S = adagrad_step(head_new, head_g, head_S, st.lr_head, st.weight_decay)
    return out, enc_new, head_new, enc_S, head_S, enc_g, head_g


# -------------------------------------------------------------------------------- AUC
def roc_auc(scores, labels) -> float:
    """Tie-aware trapezoidal ROC-AUC == sklearn ``roc_curve`` + ``auc`` as used by
    utils/eval_utils.py:21-24,139-143 (pos_label=1)."""
    import numpy as np
    s = np.asarray(scores, dtype=np.float64).ravel()
    y = np.asarray(labels).ravel().astype(np.float64)
    order = np.argsort(-s, kind="mergesort")
    s, y = s[order], y[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[idx]
    fps = 1 + idx - tps
    tps, fps = np.r_[0, tps], np.r_[0, fps]
    if tps[-1] == 0 or fps[-1] == 0:
        return float("nan")
    trap = getattr(np, "trapezoid", None) or np.trapz
    return float(trap(tps / tps[-1], fps / fps[-1]))
