"""Drop-in for the reference ``models/MultiHeadAttention.py`` (class surface: :28-30, forward :93-132).

Parameters keep the reference names (``w_qs/w_ks/w_vs/fc`` bias-free ``nn.Linear`` holders, ``layer_norm``,
``relative_position_bias_table`` + buffer ``relative_position_index``) so published checkpoints load; the
arithmetic runs in ``MHAFunction`` (GEMM -> fused attention core -> GEMM epilogue -> LayerNorm kernels).
"""
import torch
from torch import nn

from ..functional import MHAClsAssocFunction, MHAClsFunction, MHAFunction, PackedAct


def relative_position_index_3d(window_depth: int, window_size: int) -> torch.Tensor:
    """int64 [L*ws^2, L*ws^2]: token t = (d, h, w) row-major over (depth, height, width);
    entry = (dd+L-1)(2ws-1)^2 + (dh+ws-1)(2ws-1) + (dw+ws-1) — same table the reference registers (:56-73)."""
    n = window_depth * window_size * window_size
    t = torch.arange(n)
    dd = (t // (window_size * window_size)).view(-1, 1) - (t // (window_size * window_size)).view(1, -1)
    dh = ((t // window_size) % window_size).view(-1, 1) - ((t // window_size) % window_size).view(1, -1)
    dw = (t % window_size).view(-1, 1) - (t % window_size).view(1, -1)
    span = 2 * window_size - 1
    return ((dd + window_depth - 1) * span * span + (dh + window_size - 1) * span + (dw + window_size - 1)).long()


def relative_position_index_2d(window_size: int) -> torch.Tensor:
    """2-D variant (:79-89): entry = (dh+ws-1)(2ws-1) + (dw+ws-1)."""
    t = torch.arange(window_size * window_size)
    dh = (t // window_size).view(-1, 1) - (t // window_size).view(1, -1)
    dw = (t % window_size).view(-1, 1) - (t % window_size).view(1, -1)
    span = 2 * window_size - 1
    return ((dh + window_size - 1) * span + (dw + window_size - 1)).long()


class MultiHeadAttention(nn.Module):
    def __init__(self, n_head, d_model, d_k, d_v, layerNorm=False,
                 attn_dropout=0.1, fc_dropout=0.1, relative_pe=False, window_size=3,
                 window_depth=3, conv_patch=False, relative_pe_2D=False):
        super().__init__()
        self.n_head, self.d_k, self.d_v, self.d_model = n_head, d_k, d_v, d_model
        self.layerNorm_flag = layerNorm
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(d_model, n_head * d_v, bias=False)
        self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
        self.dropout = nn.Dropout(fc_dropout)            # rate holders (state-less); masks come from the HIP RNG
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.temperature = d_k ** 0.5
        self.attn_dropout = nn.Dropout(attn_dropout)
        self.relative_pe, self.relative_pe_2D = relative_pe, relative_pe_2D
        self.window_size, self.window_depth = window_size, window_depth
        if relative_pe:
            rows = (2 * window_depth - 1) * (2 * window_size - 1) ** 2
            self.relative_position_bias_table = nn.Parameter(torch.zeros(rows, n_head))
            self.register_buffer("relative_position_index", relative_position_index_3d(window_depth, window_size))
            nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        if relative_pe_2D:
            self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, n_head))
            self.register_buffer("relative_position_index", relative_position_index_2d(window_size))
            nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self._site = ""
        self._act16_out = False        # bf16 activation stream: hand the result on as a PackedAct (set per call by Encoder)
        self.cls_assoc = True          # last-layer CLS attention without materialising K / V (see functional)

    def fuse_qkv_(self):
        """Re-home w_qs / w_ks / w_vs as consecutive row blocks of ONE [2*H*dk + H*dv, d_model] buffer (the Parameters,
        their names and values are unchanged; only ``.data`` is re-pointed).  MHAFunction then runs one projection GEMM
        instead of three (X is read once), one weight-gradient GEMM and one input-gradient GEMM.  Call after the module
        sits on its final device; ``load_state_dict`` / the optimizer keep working (they update ``.data`` in place)."""
        ws = [self.w_qs.weight, self.w_ks.weight, self.w_vs.weight]
        flat = torch.empty((sum(w.shape[0] for w in ws), ws[0].shape[1]), device=ws[0].device, dtype=ws[0].dtype)
        off = 0
        with torch.no_grad():
            for w in ws:
                flat[off: off + w.shape[0]].copy_(w.data)
                w.data = flat[off: off + w.shape[0]]
                off += w.shape[0]
        return self

    def _cls_assoc_ok(self, S):
        return self.d_model % 4 == 0 and self.n_head <= 16 and S <= 128 and self.n_head * self.d_model <= 16384 and self.cls_assoc

    def cls_takes_pack(self, N, S):
        """Can ``forward_cls`` read its input as a PackedAct (functional.cls_pack_ok: lstc_cls_dot_pack and friends)?"""
        from ..functional import cls_pack_ok
        return self._cls_assoc_ok(S) and cls_pack_ok(N, S, self.n_head, self.d_model)

    def forward_cls(self, x):
        """CLS-query attention for the last encoder layer: x [N, S, d] -> [N, d] (== ``forward(x, x, x)[0][:, 0]``)."""
        cfg = dict(n_head=self.n_head, d_k=self.d_k, d_v=self.d_v, layer_norm=self.layerNorm_flag,
                   attn_dropout=self.attn_dropout.p, fc_dropout=self.dropout.p, training=self.training,
                   site=self._site)
        if isinstance(x, PackedAct):       # bf16 activation stream: Encoder checked cls_takes_pack(); the pack's bf16 view goes through autograd
            cfg.update(act_shape=x.shape)
            fn, x = MHAClsAssocFunction, x.t
        else:
            # re-associated form (no K/V projection GEMMs) whenever its alignment rules hold, else the K/V-projecting form
            fn = MHAClsAssocFunction if self._cls_assoc_ok(x.shape[1]) else MHAClsFunction
        return fn.apply(x, self.w_qs.weight, self.w_ks.weight, self.w_vs.weight, self.fc.weight,
                                    self.layer_norm.weight if self.layerNorm_flag else None,
                                    self.layer_norm.bias if self.layerNorm_flag else None,
                                    self.relative_position_bias_table if (self.relative_pe or self.relative_pe_2D) else None,
                                    cfg)

    def forward(self, q, k, v, mask=None, return_attn=False, return_attn_v=False):
        if mask is not None:
            raise NotImplementedError("attention masks are never passed on the LSTC_VAD path (SURVEY 8a A3)")
        if not (q is k and k is v):
            raise NotImplementedError("only self-attention (q is k is v) is on the LSTC_VAD path")
        has_bias = self.relative_pe or self.relative_pe_2D
        if self.relative_pe_2D and q.shape[1] - 1 != self.window_size ** 2:
            raise RuntimeError("relative_pe_2D needs window_size**2 patch tokens (models/MultiHeadAttention.py:114)")
        cfg = dict(n_head=self.n_head, d_k=self.d_k, d_v=self.d_v, layer_norm=self.layerNorm_flag,
                   attn_dropout=self.attn_dropout.p, fc_dropout=self.dropout.p, training=self.training,
                   site=self._site)
        act = isinstance(q, PackedAct)
        if act:        # bf16 activation stream (functional.PackedAct): the pack's bf16 view goes through autograd, the shape rides in cfg
            if return_attn_v:
                raise NotImplementedError("return_attn_v needs the f32 activations (Encoder keeps them when it is asked for)")
            cfg.update(act_shape=q.shape, act16_out=self._act16_out)
            shape, q = q.shape, q.t
        out, probs = MHAFunction.apply(
            q, self.w_qs.weight, self.w_ks.weight, self.w_vs.weight, self.fc.weight,
            self.layer_norm.weight if self.layerNorm_flag else None,
            self.layer_norm.bias if self.layerNorm_flag else None,
            self.relative_position_bias_table if has_bias else None,
            self.relative_position_index if has_bias else None, cfg)
        if act and self._act16_out:
            out = PackedAct(out, shape)
        if return_attn_v:
            N, S = q.shape[0], q.shape[1]
            from ..functional import gemm
            vv = gemm(q.contiguous().view(N * S, -1), self.w_vs.weight, trans_b=True)
            return out, probs, vv.view(N, S, self.n_head, self.d_v).transpose(1, 2)
        if not return_attn:
            return out, None
        return out, probs
