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


def host_dropout_keep(i, p, seed):
    """numpy restatement of the library's counter-based keep/drop rule (csrc/lstc_common.h: drop_key_mix = splitmix64's
    finaliser over the 64-bit seed, drop_hash = two-multiply avalanche of the flat element index): True where element ``i``
    is kept.  Part of the ABI contract - masks are replayed by tests."""
    m = (1 << 64) - 1
    zz = (int(seed) + 0x9E3779B97F4A7C15) & m
    zz = ((zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9) & m
    zz = ((zz ^ (zz >> 27)) * 0x94D049BB133111EB) & m
    zz ^= zz >> 31
    k0, k1, thr = zz & 0xffffffff, zz >> 32, min(int(p * 4294967296.0), 0xffffffff)
    u, lo = np.uint64, np.uint64(0xffffffff)
    h = (np.asarray(i).astype(np.uint64) ^ u(k0)) & lo
    h = (h * u(0x9E3779B1)) & lo
    h ^= h >> u(15)
    h = (h + u(k1)) & lo
    h = (h * u(0x85EBCA77)) & lo
    h ^= h >> u(13)
    h = (h * u(0xC2B2AE3D)) & lo
    h ^= h >> u(16)
    return h >= u(thr)


_BATCH_CACHE = {}


def cached_training_batch(*args, **kw):
    """``synthetic.training_batch`` with the last three results kept: the headline-size batches (2 x 403 MB from the portable
    generator, several seconds each) are asked for by five tests in a row.  Returns copies: a test may write into its arrays."""
    from lstc_vad_amd import synthetic as syn
    key = (args, tuple(sorted(kw.items())))
    hit = _BATCH_CACHE.get(key)
    if hit is None:
        hit = syn.training_batch(*args, **kw)
        while len(_BATCH_CACHE) >= 3:
            _BATCH_CACHE.pop(next(iter(_BATCH_CACHE)))
        _BATCH_CACHE[key] = hit
    return tuple(a.copy() for a in hit)
