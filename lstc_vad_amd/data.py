"""Batch source with the reference DataLoader's tensor contract and window-sampling rule.

The reference reads I3D features from HDF5 (utils/load_dataset.py:29-48) and samples ``part_num`` windows of
``part_len`` clips per video (``sample_feat`` :56-88).  No feature files ship with the reference, so for throughput work and
tests this module provides (i) ``sample_windows`` — the reference's uniform / random rule restated, and
(ii) ``SyntheticVideos`` — seeded synthetic videos (lstc_vad_amd.synthetic) served through that rule, normal/abnormal
pairs, ``drop_last`` batching, ``shuffle_keys`` per epoch.  Real feature files (HDF5) go through ``load_dataset.py`` +
``hdf5.py``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import synthetic as syn


def sample_windows(n_clips: int, part_num: int, part_len: int, sample: str = "uniform", rng=np.random) -> np.ndarray:
    """Clip indices [part_num*part_len] (utils/load_dataset.py:69-88).  uniform: one random shift ``move`` for all
    windows, starts = linspace(0, n-L, pn+1, dtype=int)[:pn] + move; random: an independent shift per window."""
    L = part_len
    starts = np.linspace(0, n_clips - L, part_num + 1, dtype=int)
    if sample == "uniform":
        span = (n_clips - L) // (part_num + 1)
        move = rng.randint(span) if span >= 1 else 0
        begins = starts[:part_num] + move
    else:
        gap = max(int(starts[1] - starts[0]), 1)
        begins = np.array([s + rng.randint(0, gap) for s in starts[:part_num]])
    return (begins[:, None] + np.arange(L)[None, :]).reshape(-1)


class SyntheticVideos:
    """``n_pairs`` normal + abnormal synthetic videos; iterating yields (norm_feats, norm_labs, abnorm_feats,
    abnorm_labs) exactly shaped like the reference loader's batches, already on ``device``."""

    def __init__(self, n_pairs, batch_size, part_num, part_len, n_patch, d_model, device, seed=0, sample="uniform",
                 pseudo_threshold=None, min_clips=24, max_clips=160, pseudo_labels=None, rank=0, world=1):
        if batch_size % world:
            raise ValueError(f"batch_size {batch_size} pairs do not split over {world} ranks")
        self.rank, self.world = rank, world          # rank r serves pairs [r*bs/world, (r+1)*bs/world) of each global batch
        self.bs, self.pn, self.L, self.P, self.d = batch_size, part_num, part_len, n_patch, d_model
        self.device, self.seed, self.sample, self.thr = device, seed, sample, pseudo_threshold
        rs = np.random.RandomState(seed)
        need = part_len + 1
        self.lengths = rs.randint(max(min_clips, need), max(max_clips, need + 1) + 1, size=(2, n_pairs))
        self.pseudo = pseudo_labels          # {"syn_abnormal_<i>.npy": [n_clips, 1]} as written by the generators
        self.n_pairs = n_pairs
        self.order = np.arange(n_pairs)
        self.rng = np.random.RandomState(seed + 1)

    def shuffle_keys(self):                      # reference: dataset.shuffle_keys() after every epoch
        self.rng.shuffle(self.order)

    def __len__(self):
        return len(self.order) // self.bs        # drop_last=True

    def _video(self, kind, vid, n=None):
        n = int(self.lengths[kind, vid]) if n is None else int(n)
        g = torch.Generator(device=self.device).manual_seed(self.seed * 100003 + kind * 50021 + vid)
        feats = 0.5 * torch.relu(torch.randn(n, self.P, self.d, device=self.device, generator=g))
        if kind == 1:    # abnormal videos carry a brighter anomalous stretch so a model can learn something
            a, b = n // 3, n // 3 + max(n // 4, 1)
            feats[a:b] += 0.25
            labs = torch.zeros(n, 1, device=self.device)
            labs[a:b] = 1.0
        else:
            labs = torch.zeros(n, 1, device=self.device)
        return feats, labs

    def __iter__(self):
        for b in range(len(self)):
            ids = self.order[b * self.bs:(b + 1) * self.bs]
            out = [[], [], [], []]
            bl = self.bs // self.world
            for j, vid in enumerate(ids):
                for kind in (0, 1):
                    n = int(self.lengths[kind, int(vid)])
                    w = sample_windows(n, self.pn, self.L, self.sample, self.rng)     # every rank draws every window
                    if not (self.rank * bl <= j < (self.rank + 1) * bl):
                        continue
                    feats, labs = self._video(kind, int(vid))
                    idx = torch.from_numpy(w).to(self.device)
                    f = feats[idx]
                    if kind == 0:
                        l = torch.zeros(idx.numel(), 1, device=self.device)
                    elif self.pseudo is not None:                               # labels from a generator file (:64-67)
                        pl = np.asarray(self.pseudo[self.key(1, int(vid))], np.float32).reshape(-1, 1)
                        if pl.shape[0] != feats.shape[0]:
                            raise ValueError(f"pseudo labels for {self.key(1, int(vid))} cover {pl.shape[0]} clips, video has "
                                             f"{feats.shape[0]} (generated from a different --seed / --synthetic_pairs?)")
                        l = torch.from_numpy(pl).to(self.device)[idx]
                    elif self.thr is None:
                        l = torch.ones(idx.numel(), 1, device=self.device)      # no pseudo labels: ones (:59-63)
                    else:
                        u = labs[idx] * 0.9 + 0.05
                        l = torch.where(u > self.thr, u, torch.zeros_like(u))   # thresholded pseudo scores
                    out[2 * kind].append(f)
                    out[2 * kind + 1].append(l)
            yield tuple(torch.stack(x) for x in out)

    @staticmethod
    def key(kind, vid):
        return f"syn_{'abnormal' if kind else 'normal'}_{vid}.npy"

    def train_videos(self):
        """(key, features [n_clips, P, d]) of every training video — what the pseudo-label generators iterate."""
        for vid in range(self.n_pairs):
            for kind in (0, 1):
                yield self.key(kind, vid), self._video(kind, vid)[0]

    def train_videos_labelled(self):
        """(features, per-clip 0/1 labels) of every training video - the train-AUC pass of the SHT / UBnormal scripts."""
        for vid in range(self.n_pairs):
            for kind in (0, 1):
                yield self._video(kind, vid)

    def test_videos(self, n_videos=8):
        """(features [n_clips, P, d], per-clip 0/1 labels) for evaluation."""
        rs = np.random.RandomState(self.seed + 7)
        for v in range(n_videos):
            yield self._video(v % 2, 10_000 + v, rs.randint(self.L + 1, 64))
