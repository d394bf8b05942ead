"""Training-pair datasets and test loaders with the reference's names, arguments and sampling behaviour.

Reference: utils/load_dataset.py.  Every class there is the same machine with four knobs (list-file dialect, eager or
per-item feature reads, ten-crop reshaping, UCF's short-video repeat); here that machine is ``_PairSource`` and the
public classes only set the knobs.  What is kept exactly, because it decides which clips a step sees:

* the order and arguments of every ``np.random`` / ``random`` call (``permutation`` x2 in ``shuffle_keys``; per video
  one ``randint(span)`` for ``sample='uniform'`` only when ``span >= 1``, or one ``randint(0, gap, [part_num + 1])``
  for ``'random'`` only when ``gap != 0``; ten-crop draws ``random.randint(0, 9)`` per item, UCF ``crop_return`` per
  video) - with the same seed the batches are bit-identical to the reference's (tests/test_pipeline_host.py against
  fixtures produced by the real classes);
* labels: zeros / ones without pseudo labels (:59-63), the last column of 2-column pseudo labels (:66-67);
* ``n_patch`` slicing at return time (:101-106), ten-crop variants returning ``crop_i`` as a fifth item (:229-232);
* UCF: a video with ``n_clips <= part_len`` is repeated x2 along time before sampling (:417-418).

Feature files go through ``lstc_vad_amd.archive.FeatureArchive`` (HDF5 through the package's own reader ``lstc_vad_amd.hdf5``, ``.npz``, or a directory of ``.npy`` files).
``ResidentPairs`` is the MI355X-first way to serve the same items: the whole feature set lives in HBM (SHT train is
~2.4 GB, UCF-Crime ~90 GB of fp32 I3D features - both fit in 288 GB) and a batch is one device-side row gather
(``lstc_gather_rows``) driven by the host-side window indices, so no feature bytes cross PCIe after start-up.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from .archive import FeatureArchive

_SHT, _UCF, _UBN = "sht", "ucf", "ubnormal"


def _parse_train_list(dialect: str, path: str):
    """-> (normal keys, abnormal keys, {key: n_frames}) in file order.  SHT lines are ``name,label`` (:37-40), UBnormal
    ``name,...`` with the class in the name (:539-540), UCF ``Dir/name.mp4 n_frames ...`` with ``Normal_`` prefix (:392-398)."""
    normal, abnormal, frames = [], [], {}
    for line in open(path, "r").readlines():
        if dialect == _SHT:
            parts = line.strip().split(",")
            key, is_normal = parts[0], int(parts[-1]) == 0
        elif dialect == _UBN:
            key = line.strip().split(",")[0]
            is_normal = key.split("_")[0] == "normal"
        else:
            fields = line.strip().split(" ")
            key = fields[0].split("/")[-1].split(".")[0]
            frames[key] = int(fields[1])
            is_normal = key.split("_")[0] == "Normal"
        (normal if is_normal else abnormal).append(key)
    return normal, abnormal, frames


def window_indices(feat_len: int, part_num: int, part_len: int, sample: str) -> np.ndarray:
    """Clip indices ``[part_num * part_len]`` of one video; consumes the global ``np.random`` stream exactly like
    ``sample_feat`` (utils/load_dataset.py:69-88)."""
    starts = np.linspace(0, feat_len - part_len, num=part_num + 1, dtype=int)
    if sample == "uniform":
        span = (feat_len - part_len) // (part_num + 1)
        shift = np.random.randint(span) if span >= 1 else 0
    else:
        gap = starts[1] - starts[0]
        shift = np.random.randint(0, gap, [part_num + 1]) if gap != 0 else 0
    begins = (starts + shift)[:part_num]
    return (begins[:, None] + np.arange(part_len, dtype=int)[None, :]).reshape(-1)


class _PairSource(torch.utils.data.Dataset):
    dialect = _SHT
    lazy = False            # read features per item instead of at construction
    ten_crop = False        # features stored as [n_clips*10*n_patch, d] -> [n_clips, 10, n_patch, d], one crop per item
    strip_key_suffix = False  # SH ten-crop looks pseudo labels up without the ".npy" suffix (:219-220)

    def __init__(self, part_num, part_len, h5_path, train_txt, n_patch, sample, pseudo_labels_path=None, d_model=None):
        self.part_num, self.part_len, self.h5_path, self.train_txt = part_num, part_len, h5_path, train_txt
        self.n_patch, self.sample, self.d_model = n_patch, sample, d_model
        if pseudo_labels_path is not None:
            if not os.path.exists(pseudo_labels_path):
                print("Can NOT open the pseudo labels file!")
                raise SystemExit(-1)                       # reference: exit(-1) (:22-23)
            print("Pseudo labels load successful.")
            self.pseudo_labels = np.load(pseudo_labels_path, allow_pickle=True).tolist()
        else:
            self.pseudo_labels = None
        self.load_feat()
        self.shuffle_keys()

    # -- construction ------------------------------------------------------------------------------------------
    def load_feat(self):
        normal, abnormal, self.video_number_frames = _parse_train_list(self.dialect, self.train_txt)
        self.norm_keys = [k + ".npy" for k in normal]
        self.abnorm_keys = [k + ".npy" for k in abnormal]
        if self.lazy:
            self.norm_feats, self.abnorm_feats = list(self.norm_keys), list(self.abnorm_keys)
        else:
            with FeatureArchive(self.h5_path) as arc:
                self.norm_feats = [self._shape(np.asarray(arc[k])) for k in self.norm_keys]
                self.abnorm_feats = [self._shape(np.asarray(arc[k])) for k in self.abnorm_keys]

    def _shape(self, feat):
        return feat.reshape((-1, 10, self.n_patch, self.d_model)) if self.ten_crop else feat

    def __len__(self):
        return min(len(self.norm_feats), len(self.abnorm_feats))

    def shuffle_keys(self):
        self.norm_iters = np.random.permutation(len(self.norm_feats))
        self.abnorm_iters = np.random.permutation(len(self.abnorm_feats))

    # -- one item ----------------------------------------------------------------------------------------------
    def _fetch(self, entry):
        if not self.lazy:
            return entry
        with FeatureArchive(self.h5_path) as arc:
            return np.asarray(arc[entry])

    def _labels_for(self, feat_len, labs, vid_type):
        if labs is None:
            return (np.zeros if vid_type == "Normal" else np.ones)([feat_len, 1], dtype=np.float32)
        if len(labs.shape) == 2 and labs.shape[-1] == 2:
            return labs[:, -1]
        return labs

    def sample_feat(self, feat, labs, vid_type="Normal"):
        feat = self._fetch(feat)
        labs = self._labels_for(feat.shape[0], labs, vid_type)
        idx = window_indices(feat.shape[0], self.part_num, self.part_len, self.sample)
        return feat[idx, :], labs[idx]

    def _pseudo(self, key):
        if self.pseudo_labels is None:
            return None
        return self.pseudo_labels[key[:-4] if self.strip_key_suffix else key]

    def item_plan(self, item):
        """(normal entry, normal labels, abnormal entry, abnormal labels) before any sampling."""
        ni, ai = self.norm_iters[item], self.abnorm_iters[item]
        return (self.norm_feats[ni], self._pseudo(self.norm_keys[ni]),
                self.abnorm_feats[ai], self._pseudo(self.abnorm_keys[ai]))

    def __getitem__(self, item):
        nf, nl, af, al = self.item_plan(item)
        if self.ten_crop:
            crop_i = random.randint(0, 9)
            nf, af = nf[:, crop_i, :, :], af[:, crop_i, :, :]
        norm_feat, norm_labs = self.sample_feat(nf, nl, vid_type="Normal")
        abnorm_feat, abnorm_labs = self.sample_feat(af, al, vid_type="Abnormal")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
        if self.ten_crop:
            out = (t(norm_feat), t(norm_labs), t(abnorm_feat), t(abnorm_labs))
            return out if self.n_patch == 1 else out + (crop_i,)
        if self.n_patch == 1:
            return t(norm_feat), t(norm_labs), t(abnorm_feat), t(abnorm_labs)
        return t(norm_feat[:, :self.n_patch, :]), t(norm_labs), t(abnorm_feat[:, :self.n_patch, :]), t(abnorm_labs)


class SH_Train_Origin_Dataset(_PairSource):
    """utils/load_dataset.py:9-106."""


class SH_Train_Origin_Dataset_MutualTraining(_PairSource):
    """utils/load_dataset.py:234-336 - features are re-read from the archive for every item."""
    lazy = True


class SH_Train_Origin_Dataset_tenCrop(_PairSource):
    """utils/load_dataset.py:134-232."""
    ten_crop = True
    strip_key_suffix = True

    def __init__(self, part_num, part_len, h5_path, train_txt, n_patch, sample, d_model, pseudo_labels_path=None):
        super().__init__(part_num, part_len, h5_path, train_txt, n_patch, sample, pseudo_labels_path, d_model)


class UBnormal_Train_Origin_Dataset(_PairSource):
    """utils/load_dataset.py:512-604."""
    dialect = _UBN


class UBnormal_Train_Origin_Dataset_tenCrop(_PairSource):
    """utils/load_dataset.py:631-729."""
    dialect = _UBN
    ten_crop = True

    def __init__(self, part_num, part_len, h5_path, train_txt, n_patch, sample, d_model, pseudo_labels_path=None):
        super().__init__(part_num, part_len, h5_path, train_txt, n_patch, sample, pseudo_labels_path, d_model)


class UCF_Train_Origin_Dataset(_PairSource):
    """utils/load_dataset.py:364-463.  Attribute names follow the reference (``norm_video_names_list`` ...)."""
    dialect = _UCF
    lazy = True

    def __init__(self, part_num, part_len, frames_per_clip, h5_path, train_txt, n_patch, sample, pseudo_labels_path=None,
                 d_model=4096, crop_return=False):
        self.frames_per_clip, self.crop_return = frames_per_clip, crop_return
        super().__init__(part_num, part_len, h5_path, train_txt, n_patch, sample, pseudo_labels_path, d_model)
        self.norm_video_names_list = [k[:-4] for k in self.norm_keys]
        self.abnorm_video_names_list = [k[:-4] for k in self.abnorm_keys]

    def _fetch(self, entry):
        feat = super()._fetch(entry)
        if self.crop_return:
            feat = feat.reshape((-1, 10, self.n_patch, self.d_model))[:, random.randint(0, 9), :, :]
        if feat.shape[0] <= self.part_len:
            feat = np.repeat(feat, 2, axis=0)
        return feat


# -- test-set loaders ------------------------------------------------------------------------------------------------

def _test_set(dialect, txt_path, mask_dir, h5_file, return_names, reshape=None):
    feats, labels, annos, names = [], [], [], []
    with FeatureArchive(h5_file) as arc:
        for line in open(txt_path, "r").readlines():
            parts = line.strip().split(",")
            key = parts[0]
            abnormal = (parts[1] == "1") if dialect == _SHT else (key.split("_")[0] == "abnormal")
            n_frames = parts[-1] if dialect == _SHT else parts[1]
            feat = np.asarray(arc[key + ".npy"])
            feats.append(feat.reshape(reshape) if reshape else feat)
            annos.append(np.load(os.path.join(mask_dir, key + ".npy")) if abnormal else np.zeros(int(n_frames)))
            labels.append("Abnormal" if abnormal else "Normal")
            names.append(key)
    return (feats, labels, annos, names) if return_names else (feats, labels, annos)


def shanghaitech_test(txt_path, mask_dir, h5_file, return_names=False):
    """utils/load_dataset.py:108-132: lines ``name,label,...,n_frames``; frame masks ``<mask_dir>/<name>.npy``."""
    return _test_set(_SHT, txt_path, mask_dir, h5_file, return_names)


def shanghaitech_test_tenCrop(txt_path, mask_dir, h5_file, n_patch, d_model, return_names=False):
    """utils/load_dataset.py:338-362."""
    return _test_set(_SHT, txt_path, mask_dir, h5_file, return_names, (-1, 10, n_patch, d_model))


def UBnormal_test(txt_path, mask_dir, h5_file, return_names=False):
    """utils/load_dataset.py:606-629: lines ``name,n_frames``; class from the ``abnormal_`` / ``normal_`` prefix."""
    return _test_set(_UBN, txt_path, mask_dir, h5_file, return_names)


def UBnormal_test_tenCrop(txt_path, mask_dir, h5_file, n_patch, d_model, return_names=False):
    """utils/load_dataset.py:731-755."""
    return _test_set(_UBN, txt_path, mask_dir, h5_file, return_names, (-1, 10, n_patch, d_model))


def _ucf_line(line):
    fields = line.strip().split(" ")
    return fields[0].split("/")[1].split(".")[0], int(fields[1]), fields


def UCF_train(line, data_h5_file_path, frames_per_clip=16, return_name=False):
    """utils/load_dataset.py:465-475."""
    key, n_frames, _ = _ucf_line(line)
    with FeatureArchive(data_h5_file_path) as arc:
        feats = np.asarray(arc[key + ".npy"])
    return (feats, n_frames, key) if return_name else (feats, n_frames)


def UCF_test(line, data_h5_file_path, gt_h5_file_path, frames_per_clip=16, return_name=False):
    """utils/load_dataset.py:477-492: ``Dir/name.mp4 n_frames label``; frame-level ground truth of abnormal videos
    comes from a second archive."""
    key, n_frames, fields = _ucf_line(line)
    with FeatureArchive(data_h5_file_path) as arc:
        feats = np.asarray(arc[key + ".npy"])
    if fields[2] == "Normal":
        anno = np.zeros(n_frames)
    else:
        with FeatureArchive(gt_h5_file_path) as gt:
            anno = np.asarray(gt[key + ".npy"])
    return (feats, anno, n_frames, key) if return_name else (feats, anno, n_frames)


UCF_test_tenCrop = UCF_test        # utils/load_dataset.py:494-510 is the same function under another name


# -- HBM-resident serving of the same items --------------------------------------------------------------------------

class WorkerStreams:
    """The RNG streams of the reference's ``DataLoader(num_workers=k, worker_init_fn=worker_init)``.

    Upstream samples windows (and ten-crop indices) INSIDE the loader's worker processes: worker ``w`` produces batches
    ``w, w + k, w + 2k, ...`` of an epoch and its ``np.random`` / ``random`` generators are seeded ``seed + w`` by
    ``worker_init`` (Train/temporal_transformer_shanghaitech.py:39-41).  The workers are re-forked by every ``for ... in
    dataloader``, i.e. re-seeded at every epoch, and the parent's generators - which ``shuffle_keys`` consumes - never see a
    sampling draw.  ``with streams.batch(b):`` installs the stream of the worker that owns batch ``b`` of the current epoch
    and puts the parent's generators back afterwards; ``begin_epoch()`` re-seeds.  ``k = 0`` (``num_workers=0``) leaves
    everything on the caller's generators, the order the dataset fixtures pin."""

    def __init__(self, k: int, seed: int):
        self.k, self.seed = int(k), int(seed)
        self._np, self._py = [None] * self.k, [None] * self.k
        self.begin_epoch()

    def begin_epoch(self):
        if self.k <= 0:
            return
        keep_np, keep_py = np.random.get_state(), random.getstate()
        for w in range(self.k):
            np.random.seed(self.seed + w); random.seed(self.seed + w)
            self._np[w], self._py[w] = np.random.get_state(), random.getstate()
        np.random.set_state(keep_np); random.setstate(keep_py)

    def batch(self, b: int):
        return _WorkerScope(self, b % self.k if self.k > 0 else -1)


class _WorkerScope:
    def __init__(self, streams, w):
        self.s, self.w = streams, w

    def __enter__(self):
        if self.w >= 0:
            self.keep = (np.random.get_state(), random.getstate())
            np.random.set_state(self.s._np[self.w]); random.setstate(self.s._py[self.w])
        return self

    def __exit__(self, *exc):
        if self.w >= 0:
            self.s._np[self.w], self.s._py[self.w] = np.random.get_state(), random.getstate()
            np.random.set_state(self.keep[0]); random.setstate(self.keep[1])
        return False


def shard_plan(ds, offsets, n_norm, b, bs, rank=0, world=1, crops=1):
    """Window indices + labels of global batch ``b`` (``bs`` pairs) for rank ``rank`` of ``world``: pairs
    ``[rank*bs/world, (rank+1)*bs/world)``.  The sampler runs for EVERY pair (same ``np.random`` consumption on every rank
    as in the single-process run), so the ranks' shards are disjoint and their union is the single-process batch.  A video's
    clip count is the length of its slice of the bank (``offsets``): lazy datasets hold only keys in ``norm_feats``.

    ``crops`` = 10 (round 6): a ten-crop bank holds clip ``c``'s crop ``k`` in row ``10 c + k`` of the video's slice.  The crop is
    drawn from Python's ``random`` exactly where the dataset's own item code draws it - ONE ``randint(0, 9)`` per item in front of
    both videos' windows (``_PairSource.__getitem__``, utils/load_dataset.py:229-232), or one per video at fetch time (UCF's
    ``crop_return``, :437-438) - on every rank, for every pair of the global batch."""
    rows = ds.part_num * ds.part_len
    bl = bs // world
    lo = rank * bl
    idx = np.empty((2, bl, rows), np.int64)
    labs = np.empty((2, bl, rows, 1), np.float32)
    per_fetch = crops > 1 and not ds.ten_crop            # UCF crop_return: the crop is drawn inside _fetch, once per video
    for j in range(bs):
        item = b * bs + j
        ni, ai = ds.norm_iters[item], ds.abnorm_iters[item]
        crop = random.randint(0, 9) if (crops > 1 and not per_fetch) else 0
        for kind, vid in ((0, ni), (1, ai)):
            slot = vid + (n_norm if kind else 0)
            if per_fetch:
                crop = random.randint(0, 9)
            n_clips = int(offsets[slot + 1] - offsets[slot]) // crops
            keys = ds.norm_keys if kind == 0 else ds.abnorm_keys
            l = ds._labels_for(n_clips, ds._pseudo(keys[vid]), "Normal" if kind == 0 else "Abnormal")
            w = window_indices(n_clips, ds.part_num, ds.part_len, ds.sample)     # consumes np.random on every rank
            if lo <= j < lo + bl:
                idx[kind, j - lo] = w * crops + crop + offsets[slot]
                labs[kind, j - lo] = np.asarray(l, np.float32).reshape(-1, 1)[w] if np.ndim(l) == 1 else l[w]
    return idx, labs


class ResidentPairs:
    """Serve a ``_PairSource``'s batches from HBM.

    All videos of the (eager, single-crop) dataset are uploaded once into ONE flat ``[total_clips, n_patch, d]`` tensor;
    a batch is formed by ``lstc_gather_rows`` from the window indices that the dataset's own sampler produces on the host
    (same ``np.random`` consumption as iterating the dataset with ``DataLoader(batch_size, drop_last=True,
    num_workers=0, shuffle=False)``), so the batches are bit-identical to the host path while no feature bytes cross
    PCIe per step.  Labels are tiny and travel with the indices.

    Round 5: the LAZY single-crop datasets too (``SH_Train_Origin_Dataset_MutualTraining`` - the co-teaching stage of BASELINE
    config 3 - and ``UCF_Train_Origin_Dataset`` without ``crop_return``).  Upstream re-reads a video from the archive for every
    item because the host cannot hold the set; 288 GB of HBM can (UCF-Crime: ~90 GB of fp32 features), and what an item holds
    does not depend on when its video was read: each video goes through the dataset's own ``_fetch`` ONCE at construction (UCF:
    videos of at most ``part_len`` clips are repeated, utils/load_dataset.py:437-438) and lands in the bank.  The host-staged path
    (``cli._HostPairs``) moved 1.2 GB per step over PCIe for the reference's MIL_CE batch: 717 ms per step against 58 ms for the
    same model on the resident feed (tools/coteach_round.py)."""

    def __init__(self, dataset: _PairSource, batch_size: int, device, rank: int = 0, world: int = 1, streams: "WorkerStreams" = None):
        """``batch_size`` = pairs of the GLOBAL batch (the reference's ``--batch_size``); under data parallelism rank ``r``
        of ``world`` serves pairs ``[r*bs/world, (r+1)*bs/world)`` of every global batch.  Every rank runs the
        sampler for the whole global batch - same ``np.random`` consumption as the single-process run - so the ranks'
        shards are disjoint and their union IS the single-process batch."""
        from .feed import ResidentBank
        if not self.serves(dataset):
            raise ValueError("ResidentPairs serves _PairSource datasets")
        if batch_size % world:
            raise ValueError(f"--batch_size {batch_size} pairs do not split over {world} ranks")
        self.ds, self.bs, self.device, self.rank, self.world = dataset, batch_size, device, rank, world
        self.streams = streams or WorkerStreams(0, 0)
        P = dataset.n_patch
        entries = dataset.norm_feats + dataset.abnorm_feats           # arrays (eager) or archive keys (lazy)
        crop_return = bool(getattr(dataset, "crop_return", False))
        # round 6: the ten-crop datasets (utils/load_dataset.py:134-232, :631-729) and UCF's crop_return (:437-438) too - all ten crops
        # of every clip are resident ([clips * 10, P, d]: ten times the single-crop bank) and the item's crop index, drawn on the host
        # where the dataset draws it, is part of the gathered row index (shard_plan)
        self.crops = 10 if (dataset.ten_crop or crop_return) else 1
        if dataset.ten_crop:
            cut = lambda f: f.reshape((-1,) + tuple(f.shape[2:]))                        # [n, 10, P, d] -> [10 n, P, d]; items keep every stored patch
        else:
            cut = (lambda f: f) if P == 1 else (lambda f: f[:, :P, :])

        def rows_of(e):
            """A video as the rows it contributes to the bank."""
            if crop_return:
                # the dataset's own _fetch draws a crop; here: the raw read, all ten crops, then the short-video repeat per clip
                f = np.asarray(_PairSource._fetch(dataset, e)).reshape((-1, 10, dataset.n_patch, dataset.d_model))
                if f.shape[0] <= dataset.part_len:
                    f = np.repeat(f, 2, axis=0)
                return f.reshape((-1, dataset.n_patch, dataset.d_model))
            return dataset._fetch(e) if dataset.lazy else e
        self.n_norm = len(dataset.norm_feats)
        if dataset.lazy:
            # two passes over the archive: clip counts first (one allocation of the bank), then one upload per video; nothing but
            # the video in flight is held on the host
            keep_np, keep_py = np.random.get_state(), random.getstate()      # _fetch of a single-crop dataset draws nothing; be sure
        lens = [int(cut(rows_of(e)).shape[0]) for e in entries] if (dataset.lazy or dataset.ten_crop) else [int(v.shape[0]) for v in entries]
        self.offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        self.row_shape = tuple(cut(rows_of(entries[0])).shape[1:])
        self.bank = torch.empty((int(self.offsets[-1]),) + self.row_shape, dtype=torch.float32, device=device)
        for o, e in zip(self.offsets[:-1], entries):         # one staged copy per video, then everything is resident
            v = cut(rows_of(e))
            self.bank[o:o + v.shape[0]].copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))
        if dataset.lazy:
            np.random.set_state(keep_np); random.setstate(keep_py)
        self.feed = ResidentBank(self.bank)

    @staticmethod
    def serves(dataset) -> bool:
        """Every ``_PairSource``: single-crop (rounds 2-5) and, since round 6, the ten-crop classes and UCF with ``crop_return``."""
        return isinstance(dataset, _PairSource)

    @staticmethod
    def bank_bytes(dataset) -> int:
        """Bytes the bank of ``dataset`` would take (fp32), from the archive's shapes alone."""
        from .archive import FeatureArchive
        P = dataset.n_patch
        if not dataset.lazy:
            return int(sum(4 * v.shape[0] * int(np.prod(v.shape[1:])) * (min(P, v.shape[1]) / v.shape[1] if v.ndim == 3 else 1.0)
                           for v in dataset.norm_feats + dataset.abnorm_feats))
        tot = 0
        with FeatureArchive(dataset.h5_path) as arc:
            for k in dataset.norm_feats + dataset.abnorm_feats:
                shp = arc.shape(k)
                n_clips = int(np.prod(shp)) // (10 * dataset.n_patch * dataset.d_model) if getattr(dataset, "crop_return", False) else shp[0]
                tot += 4 * int(np.prod(shp)) * (2 if n_clips <= dataset.part_len and hasattr(dataset, "frames_per_clip") else 1)
        return int(tot)

    def __len__(self):
        return len(self.ds) // self.bs

    def plan(self, b):
        """Host side of batch ``b``: (clip rows into the bank ``[2, bs_local, pn*L]`` int64, labels ``[2, bs_local, pn*L, 1]``)
        of THIS rank's shard.  Draws the windows of every pair of the global batch, in the dataset's own order."""
        return shard_plan(self.ds, self.offsets, self.n_norm, b, self.bs, self.rank, self.world, self.crops)

    def __iter__(self):
        self.streams.begin_epoch()                       # upstream re-forks (re-seeds) the loader's workers every epoch
        for b in range(len(self)):
            with self.streams.batch(b):
                idx, labs = self.plan(b)
            # lazy_rows (set by the training CLI): the halves are feed.LazyRows - engine.TrainStep fuses the gather into the CLS
            # concat; default: the gathered tensors (what DataLoader would have collated)
            out, labs_d = self.feed.gather(idx, labs, lazy=getattr(self, "lazy_rows", False))
            yield out[0], labs_d[0], out[1], labs_d[1]

    def shuffle_keys(self):
        self.ds.shuffle_keys()
