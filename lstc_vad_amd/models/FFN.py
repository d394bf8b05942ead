"""Drop-in for the reference ``models/FFN.py`` (:4-22): ``LN?(dropout(W2 relu(W1 x + b1) + b2) + x)``."""
from torch import nn

from ..functional import FFNFunction, PackedAct


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_in, d_hid, dropout=0.1, layerNorm=True):
        super().__init__()
        self.w_1 = nn.Linear(d_in, d_hid)
        self.w_2 = nn.Linear(d_hid, d_in)
        self.layer_norm = nn.LayerNorm(d_in, eps=1e-6)
        self.dropout = nn.Dropout(dropout)
        self.layerNorm_flag = layerNorm
        self._site = ""
        self._act16_out = False    # bf16 activation stream: hand the result on as a PackedAct (set per call by Encoder)
        self._emit_pack = True     # bf16 mode: the LayerNorm also writes the packed operand of the NEXT full layer (set by Encoder)

    def forward(self, x):
        cfg = dict(dropout=self.dropout.p, training=self.training, layer_norm=self.layerNorm_flag, site=self._site,
                   emit_pack=self._emit_pack)
        act = isinstance(x, PackedAct)
        if act:
            cfg.update(act_shape=x.shape, act16_out=self._act16_out)
            shape, x = x.shape, x.t
        out = FFNFunction.apply(x, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias,
                                self.layer_norm.weight if self.layerNorm_flag else None,
                                self.layer_norm.bias if self.layerNorm_flag else None, cfg)
        return PackedAct(out, shape) if (act and self._act16_out) else out
