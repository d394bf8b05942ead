"""Shared helpers for the parity tests: load a golden case, build oracle configs."""
import os

import numpy as np
import torch

from cases import CASES
from oracle import lstc_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    mode, enc_kw, st_kw = CASES[name]
    return z, mode, dict(enc_kw), dict(st_kw)


def sub(z, prefix, as_torch=True):
    out = {}
    for k in z.files:
        if k.startswith(prefix):
            v = z[k]
            out[k[len(prefix):]] = torch.from_numpy(np.array(v)) if as_torch else np.array(v)
    return out


def oracle_cfgs(mode, enc_kw, st_kw, dropout=0.0):
    ecfg = orc.EncoderCfg(n_layers=3, MHA_attn_dropout=dropout, MHA_fc_dropout=dropout, FFN_dropout=dropout,
                          position_dropout=dropout, **enc_kw)
    st = orc.StepCfg(mode=mode, head_dropout=dropout, **st_kw)
    return ecfg, st


def max_abs_diff(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0
