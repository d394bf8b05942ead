"""Drop-in for the reference ``models/Encoder.py`` (:4-74)."""
import torch
from torch import nn

from ..functional import ClsConcatFunction, DropoutFunction, LayerNormFunction, PackedAct, act_chain_ok, drop_producer_packs
from .EncoderLayer import EncoderLayer


class Encoder(nn.Module):
    def __init__(self, n_layers, n_head, d_k, d_v, d_model, d_inner,
                 MHA_attn_dropout=0.1, MHA_fc_dropout=0.1, MHA_layerNorm=False,
                 FFN_dropout=0.1, FFN_layerNorm=True,
                 weight_init=True, CLS_learned=False, position_dropout=0.1, position_encoding=False,
                 max_position_tokens=100, relative_pe=False, window_size=4, window_depth=3, conv_patch=False,
                 input_layerNorm=False, relative_pe_2D=False, FFN_need=True):
        super().__init__()
        self.CLS_learned = CLS_learned
        if CLS_learned:
            self.cls_token = nn.Parameter(torch.randn(1, 1, d_model))
        self.position_encoding = position_encoding
        if position_encoding:
            self.position_dropout = nn.Dropout(position_dropout)
            self.position_enc = nn.Parameter(torch.randn(1, max_position_tokens, d_model))
        self.layer_stack = nn.ModuleList([
            EncoderLayer(d_model, d_inner, n_head, d_k, d_v, MHA_attn_dropout=MHA_attn_dropout,
                         MHA_fc_dropout=MHA_fc_dropout, MHA_layerNorm=MHA_layerNorm, FFN_dropout=FFN_dropout,
                         FFN_layerNorm=FFN_layerNorm, relative_pe=relative_pe, window_size=window_size,
                         window_depth=window_depth, conv_patch=conv_patch, relative_pe_2D=relative_pe_2D,
                         FFN_need=FFN_need)
            for _ in range(n_layers)])
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.input_layerNorm = input_layerNorm
        for i, layer in enumerate(self.layer_stack):      # dropout-site names = reference module paths
            layer.slf_attn._site = f"layer_stack.{i}.slf_attn."
            layer.pos_ffn._site = f"layer_stack.{i}.pos_ffn."
        if weight_init:
            self._reset_parameters()

    def _reset_parameters(self):
        # xavier-uniform on every parameter with dim > 1, bias table / cls / pos-enc included (reference :38-41)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def used_parameters(self):
        """Parameters that receive a gradient for this configuration.  The reference creates LayerNorms it
        never calls (``layer_norm`` unless input_layerNorm, ``slf_attn.layer_norm`` unless MHA_layerNorm, ...);
        Adagrad skips them (grad is None) and so must the gradient all-reduce (SURVEY 7 'Unused parameters')."""
        skip = set()
        if not self.input_layerNorm:
            skip.update(id(p) for p in self.layer_norm.parameters())
        for layer in self.layer_stack:
            if not layer.slf_attn.layerNorm_flag:
                skip.update(id(p) for p in layer.slf_attn.layer_norm.parameters())
            if not layer.FFN_need:
                skip.update(id(p) for p in layer.pos_ffn.parameters())
            elif not layer.pos_ffn.layerNorm_flag:
                skip.update(id(p) for p in layer.pos_ffn.layer_norm.parameters())
        return [p for p in self.parameters() if id(p) not in skip]

    @staticmethod
    def _gather_spec(x):
        """``x`` = (bank [clips, P, d], clip_idx int64 [N * Lc] on the device, N, Lc): sequences still to be gathered out of an
        HBM-resident feature bank (engine.TrainStep on a feed.LazyRows batch) -> (N, S - 1, d)."""
        bank, idx, N, Lc = x
        return N, Lc * bank.shape[1], bank.shape[2]

    def _act_chain(self, enc_output, enc_output_hi, layers) -> bool:
        """bf16 mode: do ``layers`` (the full encoder layers of this call) run on the bf16 activation stream?  Only when nothing in
        front of them needs a gradient or an f32 copy (no input LayerNorm, learned CLS token or position table) and every block
        qualifies (functional.act_chain_ok)."""
        if self.input_layerNorm or self.CLS_learned or self.position_encoding:
            return False
        if not self.training:
            # evaluation keeps f32 activations between the blocks: whether the stream applies depends on the launch's
            # row count (whole 256-row pack tiles), and scoring.py batches by pool, by video or by rank shard - a video's score,
            # hence pseudo-label thresholds and AUC-based checkpoint selection, must not depend on what shares its launch
            return False
        if isinstance(enc_output, tuple):
            N, Sm1, d = self._gather_spec(enc_output)
            return act_chain_ok(N, Sm1 + 1, d, layers)
        if enc_output.requires_grad or enc_output.dtype != torch.float32 or not enc_output.is_cuda or enc_output.dim() != 3:
            return False
        N = enc_output.shape[0] + (enc_output_hi.shape[0] if enc_output_hi is not None else 0)
        return act_chain_ok(N, enc_output.shape[1] + 1, enc_output.shape[2], layers)

    def _embed(self, enc_output, enc_output_hi=None, pack_only=False):
        if isinstance(enc_output, tuple):
            # batch formation fused into the CLS concat (lstc_cls_concat_gather_fwd): the gathered batch is never written
            bank, idx, N, Lc = enc_output
            if self.input_layerNorm:
                raise RuntimeError("Encoder: a gather spec cannot feed the input LayerNorm (materialise the batch first)")
            out = ClsConcatFunction.apply(bank, self.cls_token if self.CLS_learned else None,
                                          self.position_enc if self.position_encoding else None, None, pack_only, (idx, N, Lc))
            if pack_only:
                return PackedAct(out, (N, Lc * bank.shape[1] + 1, bank.shape[2]))
            if self.position_encoding and self.training and self.position_dropout.p > 0:
                out = DropoutFunction.apply(out, self.position_dropout.p, "position_dropout")
            return out
        if pack_only:
            N = enc_output.shape[0] + (enc_output_hi.shape[0] if enc_output_hi is not None else 0)
            t = ClsConcatFunction.apply(enc_output, None, None, enc_output_hi, True)
            return PackedAct(t, (N, enc_output.shape[1] + 1, enc_output.shape[2]))
        if self.input_layerNorm:
            if enc_output_hi is not None:
                enc_output, enc_output_hi = torch.cat([enc_output, enc_output_hi], 0), None
            enc_output = LayerNormFunction.apply(enc_output, self.layer_norm.weight, self.layer_norm.bias)
        enc_output = ClsConcatFunction.apply(enc_output, self.cls_token if self.CLS_learned else None,
                                             self.position_enc if self.position_encoding else None, enc_output_hi)
        if self.position_encoding and self.training and self.position_dropout.p > 0:
            enc_output = DropoutFunction.apply(enc_output, self.position_dropout.p, "position_dropout")
        return enc_output

    def forward_cls(self, enc_output, enc_output_hi=None):
        """``forward(x)[:, 0, :]`` without computing the rows nobody reads: the train / eval loops consume only the
        CLS token of the last layer (Train/temporal_transformer_shanghaitech.py:123,
        Train/spatio_transformer_shanghaitech.py:97), so the last layer evaluates its query, output projection and
        FFN for that token alone (K/V still use every token).  Saves ~25 % of the step's FLOPs at 3 layers."""
        # enc_output_hi: optional second half of the batch (the abnormal sequences) so the caller need not cat
        n = len(self.layer_stack)
        act = self._act_chain(enc_output, enc_output_hi, list(self.layer_stack[:-1]))
        enc_output = self._embed(enc_output, enc_output_hi, pack_only=act)
        # the CLS-only layer reads the stream's pack too when its re-associated form applies (lstc_cls_dot_pack ...), else f32 rows
        cls_pack = act and isinstance(enc_output, PackedAct) and \
            self.layer_stack[-1].slf_attn.cls_takes_pack(enc_output.shape[0], enc_output.shape[1])
        for i, layer in enumerate(self.layer_stack[:-1]):
            layer.pos_ffn._emit_pack = i + 1 < n - 1          # f32 activations: the CLS-only last layer reads no packed operand
            # bf16 activation stream: every block hands a pack on
            layer.slf_attn._act16_out, layer.pos_ffn._act16_out = act, act and (i + 1 < n - 1 or cls_pack)
            enc_output = layer(enc_output)[0]
        out = self.layer_stack[-1].forward_cls(enc_output)
        drop_producer_packs()
        return out

    def forward(self, enc_output, src_mask=None, return_attn=False, return_attn_v=False):
        attn_list, v_list = [], []
        act = not return_attn_v and src_mask is None and self._act_chain(enc_output, None, list(self.layer_stack))
        enc_output = self._embed(enc_output, None, pack_only=act)
        for i, layer in enumerate(self.layer_stack):
            layer.pos_ffn._emit_pack = i + 1 < len(self.layer_stack)      # the last layer's output goes to the caller
            layer.slf_attn._act16_out, layer.pos_ffn._act16_out = act, act and i + 1 < len(self.layer_stack)
            res = layer(enc_output, slf_attn_mask=src_mask, return_attn=return_attn, return_attn_v=return_attn_v)
            enc_output = res[0]
            if return_attn or return_attn_v:
                attn_list.append(res[1])
            if return_attn_v:
                v_list.append(res[2])
        drop_producer_packs()
        if return_attn_v:
            return enc_output, attn_list, v_list
        if return_attn:
            return enc_output, attn_list
        return enc_output
