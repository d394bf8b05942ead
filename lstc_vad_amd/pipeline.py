"""File-level stages around the training step: pseudo-label generation and frame-level AUC evaluation on feature
archives, with the reference's file formats.

* ``generate_pseudo_labels``  - Train/pseudo_labels_generator_spatio.py:67-90 and
  Train/pseudo_labels_generator_temporal.py:66-146.  Output: ``{"<video>.npy": float32 [n, 1]}`` written with
  ``np.save`` (a pickled dict; read back with ``np.load(..., allow_pickle=True).tolist()``, utils/load_dataset.py:20),
  scores at or below ``threshold`` zeroed.  STN: one row per clip.  LTN (SHT / UBnormal): one row per clip, the part's
  score repeated.  LTN (UCF): 32 rows - one per bin.
* ``evaluate_auc``            - Test/evaluation_shanghaitech_ubnormal.py:69-96, Test/evaluation_UCF.py:47-88 and the
  in-loop evaluation of the train scripts: frame-level ROC-AUC over the whole test list.

Both stages keep features on the device and pool the parts of many videos (``pool_sequences``, default 2048 - the
sequence count of a headline training step) into each launch sequence (``lstc_vad_amd.scoring``); the reference scores
one part per launch.

**Sharded over ranks (round 5).**  Videos are independent, so under N ranks (``rank`` / ``world``, default: the initialised
process group) rank r scores the videos ``i % N == r`` of the list and one zero-padded SUM all-reduce of the flat
per-sequence score vector hands every rank all scores - the exchange the training loss uses for its bag maxima.  Every rank
walks the whole list on the host (shapes, masks, list lines: a video's sequence count follows from its clip count and the
flags alone), but uploads and scores only its own videos; AUC, label files and therefore the checkpoint decisions are
computed from the complete vector on every rank and are bit-identical to a single-rank pass in fp32 mode (x + 0 = x, and a
sequence's score does not depend on what shares its batch).  In the reference this evaluation runs on GPU 0 every
``inter_epoch`` epochs over all test AND train videos (Train/temporal_transformer_shanghaitech.py:151-229) while the other
GPUs idle.
"""
from __future__ import annotations

import numpy as np
import torch

from . import scoring
from .archive import FeatureArchive
from .load_dataset import UBnormal_test, UCF_test, UCF_train, _ucf_line, shanghaitech_test
from .metrics import roc_auc


def _train_keys(dataset: str, training_txt: str):
    for line in open(training_txt, "r").readlines():
        if dataset == "UCF":
            yield line, line.strip().split(" ")[0].split("/")[-1].split(".")[0]
        else:                                   # SHT "name,label", UBnormal "name,..."
            yield line, line.strip().split(",")[0]


def _dev(a, device, n_patch=None):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    if n_patch is not None and t.dim() == 3:
        t = t[:, :n_patch, :]
    return t.to(device).contiguous()


def _ranks(rank, world, group):
    """(rank, world) of a pass: explicit arguments, else the initialised process group, else a single rank."""
    if world is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(group), dist.get_world_size(group)
        return 0, 1
    return int(rank or 0), int(world)


class _ScoreBoard:
    """Scores of one pass over a video list.  Videos are registered in list order with the number of scores they contribute
    (sequences of an LTN, clips of an STN) and a ``finish(scores) -> result`` closure; the rank that owns a video (``i % world ==
    rank``) also hands over the work (sequences to pool, or a closure that scores the clips).  ``results()`` scores what is
    owned, sums the zero-padded flat vector over the ranks and runs every video's ``finish`` on every rank."""

    def __init__(self, enc, head, device, rank, world, group, pool_sequences, exchange=None):
        self.enc, self.head, self.device = enc, head, device
        self.rank, self.world, self.group, self.exchange = rank, world, group, exchange
        self.pool_sequences = pool_sequences
        self.items, self.total = [], 0             # (offset, count, finish) in list order
        self.pool, self.pooled = [], 0             # own LTN work waiting for a launch: (offset, sequences)
        self.parts = []                            # (offset, device tensor) scored segments of this rank

    def owns(self, i):
        return i % self.world == self.rank

    def _register(self, count, finish):
        off = self.total
        self.items.append((off, int(count), finish))
        self.total += int(count)
        return off

    def add_ltn(self, i, count, make_sequences, finish):
        off = self._register(count, finish)
        if self.owns(i):
            seqs = make_sequences()
            if len(seqs) != count:
                raise RuntimeError(f"video {i}: {len(seqs)} sequences, {count} announced")
            self.pool.append((off, seqs))
            self.pooled += len(seqs)
            if self.pooled >= self.pool_sequences:
                self.flush()

    def add_clips(self, i, count, score, finish):
        off = self._register(count, finish)
        if self.owns(i):
            sc = score().reshape(-1).float()
            if sc.numel() != count:
                raise RuntimeError(f"video {i}: {sc.numel()} clip scores, {count} announced")
            self.parts.append((off, sc))

    def flush(self):
        if not self.pool:
            return
        sc = scoring.ltn_sequence_scores(self.enc, self.head, [q for _, seqs in self.pool for q in seqs])
        o = 0
        for off, seqs in self.pool:
            self.parts.append((off, sc[o:o + len(seqs)]))
            o += len(seqs)
        self.pool, self.pooled = [], 0

    def results(self):
        self.flush()
        flat = torch.zeros((self.total,), device=self.device, dtype=torch.float32)
        for off, sc in self.parts:
            flat[off:off + sc.numel()] = sc
        if self.world > 1:
            if self.exchange is not None:
                self.exchange(flat)                 # tests: several ranks emulated in one process
            else:
                import torch.distributed as dist
                from .dist import all_reduce_sum
                all_reduce_sum(flat, self.group)
        host = flat.cpu().numpy()
        return [finish(host[off:off + n]) for off, n, finish in self.items]


def _threshold(scores_np, threshold):
    s = torch.from_numpy(np.ascontiguousarray(scores_np, dtype=np.float32)).reshape(-1, 1)
    return torch.where(s > threshold, s, torch.zeros_like(s)).numpy()


@torch.no_grad()
def generate_pseudo_labels(enc, head, mode, dataset, dataset_path, training_txt, threshold, part_len=1, n_patch=16,
                           d_model=None, segment_len=16, classifier_head=False, out_path=None, pool_sequences=2048,
                           rank=None, world=None, group=None, exchange=None):
    """mode 'STN' | 'LTN'; dataset 'SHT' | 'UCF' | 'UBnormal'.  Returns the dict (and writes it when ``out_path`` - on rank 0
    only under several ranks; every rank returns the complete dict)."""
    device = next(enc.parameters()).device
    d_model = d_model or enc.layer_norm.normalized_shape[0]
    rank, world = _ranks(rank, world, group)
    board = _ScoreBoard(enc, head, device, rank, world, group, pool_sequences, exchange)
    keys = []

    with FeatureArchive(dataset_path) as arc:
        for i, (line, key) in enumerate(_train_keys(dataset, training_txt)):
            keys.append(key + ".npy")
            if mode == "STN":
                board.add_clips(i, arc.shape(key + ".npy")[0],
                                lambda key=key: scoring.stn_clip_scores(enc, head, _dev(arc[key + ".npy"], device), classifier_head),
                                lambda v: _threshold(v, threshold))
                continue
            if dataset == "UCF":
                ranges = scoring.ucf_bin_ranges(part_len, rewindow=False)

                def make(line=line):
                    feats, n_frames = UCF_train(line, dataset_path, segment_len)
                    f = _dev(feats, device).view(-1, n_patch, d_model)
                    return scoring.ltn_ucf_bin_sequences(f, n_frames, part_len, segment_len, normalize=False, rewindow=False)[0]
            else:
                ranges = scoring.part_ranges(arc.shape(key + ".npy")[0], part_len)

                def make(key=key):
                    return scoring.ltn_part_sequences(_dev(arc[key + ".npy"], device), part_len, tail="short")[0]
            board.add_ltn(i, len(ranges), make,
                          lambda v, ranges=ranges: _threshold(np.concatenate([np.repeat(x, e - b) for x, (b, e) in zip(v, ranges)]),
                                                              threshold))
        rows = board.results()
    out = dict(zip(keys, rows))                   # the list-file key order of the saved dict
    if out_path and rank == 0:
        np.save(out_path, out)
    return out


def _auc_of(rows, return_frames):
    s, l = np.concatenate([r[0] for r in rows]), np.concatenate([r[1] for r in rows])
    auc = roc_auc(s, l)
    return (auc, s, l) if return_frames else auc


@torch.no_grad()
def evaluate_auc(enc, head, mode, dataset, dataset_path, testing_txt, masks, part_len, n_patch, segment_len=16,
                 return_frames=False, pool_sequences=2048, rank=None, world=None, group=None, exchange=None):
    """``masks``: directory of ``<video>.npy`` frame masks (SHT / UBnormal, ``--test_mask_dir``) or the ground-truth
    archive (UCF, ``--test_mask_path``).  LTN only scores parts; STN scores clips (in-loop evaluation of the spatio
    scripts, Train/spatio_transformer_shanghaitech.py:118-150: every clip's score x segment_len)."""
    device = next(enc.parameters()).device
    rank, world = _ranks(rank, world, group)
    board = _ScoreBoard(enc, head, device, rank, world, group, pool_sequences, exchange)

    def stn_finish(anno):
        def fin(v):
            s = np.repeat(v, segment_len)
            return s, np.asarray(anno[:s.shape[0]])
        return fin

    if dataset == "UCF":
        for i, line in enumerate(open(testing_txt, "r").readlines()):
            if board.owns(i):
                feats, anno, n_frames, _ = UCF_test(line, dataset_path, masks, segment_len, return_name=True)
            else:                                    # not ours: the list line and the ground truth say everything but the scores
                key, n_frames, fields = _ucf_line(line)
                feats = None
                if fields[2] == "Normal":
                    anno = np.zeros(n_frames)
                else:
                    with FeatureArchive(masks) as gt:
                        anno = np.asarray(gt[key + ".npy"])
            if mode != "LTN":
                if feats is None:
                    with FeatureArchive(dataset_path) as arc:
                        shp = arc.shape(_ucf_line(line)[0] + ".npy")
                        n_clips = int(np.prod(shp)) // (n_patch * shp[-1])
                else:
                    n_clips = int(np.prod(feats.shape)) // (n_patch * feats.shape[-1])

                def score(feats=feats):
                    f = _dev(feats, device)
                    return scoring.stn_clip_scores(enc, head, f.view(-1, n_patch, f.shape[-1]))
                board.add_clips(i, n_clips, score, stn_finish(anno))
                continue
            ranges = scoring.ucf_bin_ranges(part_len, rewindow=True)
            r = scoring.ucf_bin_edges(n_frames, segment_len)

            def make(feats=feats, n_frames=n_frames):
                f = _dev(feats, device)
                return scoring.ltn_ucf_bin_sequences(f.view(-1, n_patch, f.shape[-1]), n_frames, part_len, segment_len,
                                                     normalize=True, rewindow=True)[0]
            board.add_ltn(i, len(ranges), make,
                          lambda v, ranges=ranges, r=r, anno=anno: scoring.frame_scores_ucf(v, ranges, r, anno, segment_len))
    else:
        loader = shanghaitech_test if dataset in ("SHT", "MT_SHT") else UBnormal_test
        feats_l, _, annos = loader(testing_txt, masks, dataset_path)
        for i, (feats, anno) in enumerate(zip(feats_l, annos)):
            n = feats.shape[0]
            if mode != "LTN":
                board.add_clips(i, n, lambda feats=feats: scoring.stn_clip_scores(enc, head, _dev(feats, device, n_patch)),
                                stn_finish(anno))
                continue
            ranges = scoring.part_ranges(n, part_len)
            board.add_ltn(i, len(ranges),
                          lambda feats=feats: scoring.ltn_part_sequences(_dev(feats, device, n_patch), part_len, tail="rewindow")[0],
                          lambda v, ranges=ranges, anno=anno: scoring.frame_scores_sht(v, ranges, anno, segment_len))
    return _auc_of(board.results(), return_frames)


@torch.no_grad()
def evaluate_train_auc(enc, head, mode, dataset, train_archive, training_txt, mask_dir, part_len, n_patch, segment_len=16,
                       return_frames=False, pool_sequences=2048, rank=None, world=None, group=None, exchange=None):
    """Frame-level AUC over the TRAINING videos, the quantity the SHT / UBnormal train scripts select checkpoints on
    (Train/temporal_transformer_shanghaitech.py:186-229, Train/spatio_transformer_shanghaitech.py:145-172,
    Train/temporal_transformer_UBnormal.py:193-232): every training video is scored like a test video; a normal video's
    frames are labelled 0, an abnormal video's labels are the first frames of ``mask_dir + key + ".npy"`` (the upstream
    string concatenation - ``--test_mask_dir`` holds the masks of the training videos too).  Video class: second list
    field (SHT ``name,label``) or the ``abnormal`` / ``normal`` name prefix (UBnormal)."""
    device = next(enc.parameters()).device
    rank, world = _ranks(rank, world, group)
    board = _ScoreBoard(enc, head, device, rank, world, group, pool_sequences, exchange)

    def frame_labels(abnormal, key, n):
        if not abnormal:
            return np.zeros(n)
        return np.asarray(np.load(mask_dir + key + ".npy", allow_pickle=True))[:n]

    with FeatureArchive(train_archive) as arc:
        for i, line in enumerate(open(training_txt, "r").readlines()):
            fields = line.strip().split(",")
            key = fields[0]
            if dataset in ("SHT", "MT_SHT"):
                abnormal = int(fields[1]) == 1
            elif len(fields) > 1 and fields[1].strip() in ("0", "1"):
                # the upstream pass reads the second column as the 0/1 class (Train/temporal_transformer_UBnormal.py:198)
                abnormal = int(fields[1]) == 1
            else:
                # the published UBnormal list holds a FRAME COUNT there (train_video_names_frames.txt; upstream then looks for
                # a mask of every video and stops) - the class of such a line comes from the file name
                abnormal = not key.startswith("normal")
            n = arc.shape(key + ".npy")[0]
            if mode != "LTN":
                def fin(v, abnormal=abnormal, key=key):
                    s = np.repeat(v, segment_len)
                    return s, frame_labels(abnormal, key, s.shape[0])
                board.add_clips(i, n, lambda key=key: scoring.stn_clip_scores(enc, head, _dev(arc[key + ".npy"], device, n_patch)), fin)
                continue
            ranges = scoring.part_ranges(n, part_len)

            def expand(v, ranges=ranges, abnormal=abnormal, key=key):
                sfr = np.concatenate([np.full((e - b) * segment_len, float(x), np.float32) for x, (b, e) in zip(v, ranges)])
                return sfr, frame_labels(abnormal, key, sfr.shape[0])
            board.add_ltn(i, len(ranges),
                          lambda key=key: scoring.ltn_part_sequences(_dev(arc[key + ".npy"], device, n_patch), part_len, tail="rewindow")[0],
                          expand)
        rows = board.results()
    return _auc_of(rows, return_frames)
