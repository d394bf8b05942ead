"""Deterministic synthetic data for the LSTC_VAD training hot path.

The reference trains on pre-extracted I3D snippet features read from HDF5
(/root/reference utils/load_dataset.py:90-106 returns, per item, a normal and an
abnormal video sampled to ``[part_num*part_len, n_patch, d_model]`` plus
``[part_num*part_len, 1]`` labels).  No feature files ship with the reference and
there is no network, so every test / bench in this repo uses the generator below.

It is a counter-based generator built only from integer arithmetic and exact
float additions (splitmix64 -> 24-bit uniforms -> Irwin-Hall "normal"), so the
same (seed, stream, shape) gives bit-identical arrays on every machine and numpy
build: no libm transcendental is involved.  Both the oracle side (tests) and the
HIP side regenerate inputs / weights from it.
"""
from __future__ import annotations

import numpy as np

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wraps mod 2**64)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK64
        z = z ^ (z >> np.uint64(31))
    return z


_CHUNK = 1 << 24        # counters per pass: bounds the uint64 / float64 temporaries of the headline-size batches (805 MB each) to ~1 GB


def _stream_base(seed: int, stream: int) -> np.uint64:
    with np.errstate(over="ignore"):
        return _splitmix64(np.array([(seed * 0x1000003 + stream * 0x10001 + 0x5bd1e995) & 0xFFFFFFFFFFFFFFFF],
                                    dtype=np.uint64))[0]


def _uniform_at(lo: int, hi: int, base) -> np.ndarray:
    """float64 uniforms of the counters lo .. hi-1 of a stream (24 random bits each)."""
    with np.errstate(over="ignore"):
        ctr = (np.arange(lo, hi, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base) & _MASK64
    bits = _splitmix64(ctr) >> np.uint64(40)            # top 24 bits
    return bits.astype(np.float64) * (1.0 / 16777216.0)


def uniform(shape, seed: int, stream: int = 0) -> np.ndarray:
    """float32 uniforms in [0, 1) with 24 random bits each; exact and portable."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    base = _stream_base(seed, stream)
    out = np.empty(n, np.float32)
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        out[lo:hi] = _uniform_at(lo, hi, base).astype(np.float32)
    return out.reshape(shape)


def normal_range(n: int, lo: int, hi: int, seed: int, stream: int = 0) -> np.ndarray:
    """Elements lo .. hi-1 of ``normal((n,), seed, stream)`` without forming the rest (a rank's shard of a global batch):
    element i adds the uniforms of the counters i, n + i, 2 n + i, 3 n + i."""
    base = _stream_base(seed, stream)
    out = np.empty(hi - lo, np.float32)
    for a in range(lo, hi, _CHUNK):
        b = min(hi, a + _CHUNK)
        # the float32 rounding of each uniform is part of the definition (uniform() returns float32)
        u = [_uniform_at(j * n + a, j * n + b, base).astype(np.float32).astype(np.float64) for j in range(4)]
        s = (u[0] + u[1]) + (u[2] + u[3])
        out[a - lo:b - lo] = ((s - 2.0) * 1.7320508075688772).astype(np.float32)
    return out


def normal(shape, seed: int, stream: int = 0) -> np.ndarray:
    """Approximately N(0,1) float32 (sum of 4 uniforms, centred, scaled by sqrt(3))."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    return normal_range(n, 0, n, seed, stream).reshape(shape)


def features(shape, seed: int, stream: int = 0) -> np.ndarray:
    """I3D-like snippet features: post-ReLU, non-negative, ``0.5*relu(N(0,1))``."""
    x = normal(shape, seed, stream)
    return (0.5 * np.maximum(x, 0.0)).astype(np.float32)


def features_rows(shape, lo: int, hi: int, seed: int, stream: int = 0) -> np.ndarray:
    """Rows lo .. hi-1 (first axis) of ``features(shape, seed, stream)``: what one rank of a data-parallel job holds of the batch."""
    shape = tuple(shape)
    row = int(np.prod(shape[1:]))
    x = normal_range(shape[0] * row, lo * row, hi * row, seed, stream)
    return (0.5 * np.maximum(x, 0.0)).astype(np.float32).reshape((hi - lo,) + shape[1:])


def pseudo_labels(shape, seed: int, threshold: float, stream: int = 0) -> np.ndarray:
    """Pseudo labels with the reference rule ``where(score > thr, score, 0)``
    (/root/reference Train/pseudo_labels_generator_temporal.py:139-140)."""
    u = uniform(shape, seed, stream)
    return np.where(u > np.float32(threshold), u, np.float32(0.0)).astype(np.float32)


def xavier_uniform(shape, seed: int, stream: int = 0) -> np.ndarray:
    """Xavier-uniform-shaped weights (bound = sqrt(6/(fan_in+fan_out))) from the
    portable generator; used for parity runs so both sides agree without torch RNG."""
    shape = tuple(shape)
    if len(shape) < 2:
        fan_in = fan_out = shape[0] if shape else 1
    else:
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
    bound = float(np.sqrt(6.0 / float(fan_in + fan_out)))
    u = uniform(shape, seed, stream).astype(np.float64)
    return ((2.0 * u - 1.0) * bound).astype(np.float32)


def small_uniform(shape, seed: int, stream: int = 0, scale: float = 0.05) -> np.ndarray:
    """Small symmetric values for biases / LayerNorm offsets / bias tables."""
    u = uniform(shape, seed, stream).astype(np.float64)
    return ((2.0 * u - 1.0) * scale).astype(np.float32)


def training_batch(bs: int, part_num: int, part_len: int, n_patch: int, d_model: int,
                   seed: int = 0, with_pseudo: bool = True, threshold: float = 0.9):
    """One step's batch with the reference DataLoader contract:
    ``norm_feats, abnorm_feats`` float32 ``[bs, part_num*part_len, n_patch, d_model]``,
    ``norm_labs`` zeros and ``abnorm_labs`` pseudo labels (or ones) ``[bs, part_num*part_len, 1]``."""
    T = part_num * part_len
    nf = features((bs, T, n_patch, d_model), seed, stream=1)
    af = features((bs, T, n_patch, d_model), seed, stream=2)
    nl = np.zeros((bs, T, 1), np.float32)
    if with_pseudo:
        al = pseudo_labels((bs, T, 1), seed, threshold, stream=3)
    else:
        al = np.ones((bs, T, 1), np.float32)
    return nf, nl, af, al
