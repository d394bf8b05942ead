"""Drop-in for the reference ``models/EncoderLayer.py`` (:4-30): self-attention then (optionally) the FFN."""
from torch import nn

from .FFN import PositionwiseFeedForward
from .MultiHeadAttention import MultiHeadAttention


class EncoderLayer(nn.Module):
    def __init__(self, d_model, d_inner, n_head, d_k, d_v, MHA_attn_dropout=0.1, MHA_fc_dropout=0.1,
                 MHA_layerNorm=False, FFN_dropout=0.1, FFN_layerNorm=True, return_attn=False,
                 relative_pe=False, window_size=4, window_depth=3, conv_patch=False,
                 relative_pe_2D=False, FFN_need=True):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, layerNorm=MHA_layerNorm,
                                           attn_dropout=MHA_attn_dropout, fc_dropout=MHA_fc_dropout,
                                           relative_pe=relative_pe, window_size=window_size,
                                           window_depth=window_depth, conv_patch=conv_patch,
                                           relative_pe_2D=relative_pe_2D)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=FFN_dropout, layerNorm=FFN_layerNorm)
        self.FFN_need = FFN_need

    def forward(self, enc_input, slf_attn_mask=None, return_attn=False, return_attn_v=False):
        res = self.slf_attn(enc_input, enc_input, enc_input, mask=slf_attn_mask, return_attn=return_attn,
                            return_attn_v=return_attn_v)
        enc_output = self.pos_ffn(res[0]) if self.FFN_need else res[0]
        return (enc_output,) + tuple(res[1:])

    def forward_cls(self, enc_input):
        """Last-layer shortcut: returns only the CLS row [N, d]; attention output projection and FFN run on N rows
        instead of N*S (the rest of the layer's output is never read by the train loops)."""
        out = self.slf_attn.forward_cls(enc_input)
        return self.pos_ffn(out) if self.FFN_need else out
