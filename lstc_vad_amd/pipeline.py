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
"""
from __future__ import annotations

import numpy as np
import torch

from . import scoring
from .archive import FeatureArchive
from .load_dataset import UBnormal_test, UCF_test, UCF_train, shanghaitech_test
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


@torch.no_grad()
def generate_pseudo_labels(enc, head, mode, dataset, dataset_path, training_txt, threshold, part_len=1, n_patch=16,
                           d_model=None, segment_len=16, classifier_head=False, out_path=None, pool_sequences=2048):
    """mode 'STN' | 'LTN'; dataset 'SHT' | 'UCF' | 'UBnormal'.  Returns the dict (and writes it when ``out_path``)."""
    device = next(enc.parameters()).device
    d_model = d_model or enc.layer_norm.normalized_shape[0]
    out, pool = {}, []            # pool: (key, sequences, ranges) of videos waiting to be scored together

    def flush():
        if not pool:
            return
        sc = scoring.ltn_sequence_scores(enc, head, [q for _, seqs, _ in pool for q in seqs])
        off = 0
        for key, seqs, ranges in pool:
            v = sc[off:off + len(seqs)]; off += len(seqs)
            s = torch.cat([x.repeat(e - b) for x, (b, e) in zip(v, ranges)]).reshape(-1, 1)
            out[key + ".npy"] = torch.where(s > threshold, s, torch.zeros_like(s)).cpu().numpy()
        pool.clear()

    with FeatureArchive(dataset_path) as arc:
        for line, key in _train_keys(dataset, training_txt):
            if mode == "STN":
                s = scoring.stn_clip_scores(enc, head, _dev(arc[key + ".npy"], device), classifier_head).reshape(-1, 1)
                out[key + ".npy"] = torch.where(s > threshold, s, torch.zeros_like(s)).cpu().numpy()
                continue
            if dataset == "UCF":
                feats, n_frames = UCF_train(line, dataset_path, segment_len)
                f = _dev(feats, device).view(-1, n_patch, d_model)
                seqs, ranges, _ = scoring.ltn_ucf_bin_sequences(f, n_frames, part_len, segment_len, normalize=False,
                                                                rewindow=False)
            else:
                seqs, ranges = scoring.ltn_part_sequences(_dev(arc[key + ".npy"], device), part_len, tail="short")
            out[key + ".npy"] = None              # keep the list-file key order of the saved dict
            pool.append((key, seqs, ranges))
            if sum(len(p[1]) for p in pool) >= pool_sequences:
                flush()
        flush()
    if out_path:
        np.save(out_path, out)
    return out


@torch.no_grad()
def evaluate_auc(enc, head, mode, dataset, dataset_path, testing_txt, masks, part_len, n_patch, segment_len=16,
                 return_frames=False, pool_sequences=2048):
    """``masks``: directory of ``<video>.npy`` frame masks (SHT / UBnormal, ``--test_mask_dir``) or the ground-truth
    archive (UCF, ``--test_mask_path``).  LTN only scores parts; STN scores clips (in-loop evaluation of the spatio
    scripts, Train/spatio_transformer_shanghaitech.py:118-150: every clip's score x segment_len)."""
    device = next(enc.parameters()).device
    scores, labels, pool = [], [], []          # pool: (slot in scores/labels, sequences, expand-to-frames closure)

    def flush():
        if not pool:
            return
        sc = scoring.ltn_sequence_scores(enc, head, [q for _, seqs, _ in pool for q in seqs]).cpu().numpy()
        off = 0
        for slot, seqs, expand in pool:
            scores[slot], labels[slot] = expand(sc[off:off + len(seqs)]); off += len(seqs)
        pool.clear()

    def stn(f, anno):
        sc = scoring.stn_clip_scores(enc, head, f).reshape(-1).cpu().numpy()
        s = np.repeat(sc, segment_len)
        scores.append(s); labels.append(np.asarray(anno[:s.shape[0]]))

    if dataset == "UCF":
        for line in open(testing_txt, "r").readlines():
            feats, anno, n_frames, _ = UCF_test(line, dataset_path, masks, segment_len, return_name=True)
            f = _dev(feats, device)
            f = f.view(-1, n_patch, f.shape[-1])
            if mode != "LTN":
                stn(f, anno); continue
            seqs, ranges, r = scoring.ltn_ucf_bin_sequences(f, n_frames, part_len, segment_len, normalize=True, rewindow=True)
            scores.append(None); labels.append(None)
            pool.append((len(scores) - 1, seqs,
                         lambda v, ranges=ranges, r=r, anno=anno: scoring.frame_scores_ucf(v, ranges, r, anno, segment_len)))
            if sum(len(p[1]) for p in pool) >= pool_sequences:
                flush()
    else:
        loader = shanghaitech_test if dataset in ("SHT", "MT_SHT") else UBnormal_test
        feats_l, _, annos = loader(testing_txt, masks, dataset_path)
        for feats, anno in zip(feats_l, annos):
            f = _dev(feats, device, n_patch)
            if mode != "LTN":
                stn(f, anno); continue
            seqs, ranges = scoring.ltn_part_sequences(f, part_len, tail="rewindow")
            scores.append(None); labels.append(None)
            pool.append((len(scores) - 1, seqs,
                         lambda v, ranges=ranges, anno=anno: scoring.frame_scores_sht(v, ranges, anno, segment_len)))
            if sum(len(p[1]) for p in pool) >= pool_sequences:
                flush()
    flush()
    s, l = np.concatenate(scores), np.concatenate(labels)
    auc = roc_auc(s, l)
    return (auc, s, l) if return_frames else auc


@torch.no_grad()
def evaluate_train_auc(enc, head, mode, dataset, train_archive, training_txt, mask_dir, part_len, n_patch, segment_len=16,
                       return_frames=False, pool_sequences=2048):
    """Frame-level AUC over the TRAINING videos, the quantity the SHT / UBnormal train scripts select checkpoints on
    (Train/temporal_transformer_shanghaitech.py:186-229, Train/spatio_transformer_shanghaitech.py:145-172,
    Train/temporal_transformer_UBnormal.py:193-232): every training video is scored like a test video; a normal video's
    frames are labelled 0, an abnormal video's labels are the first frames of ``mask_dir + key + ".npy"`` (the upstream
    string concatenation - ``--test_mask_dir`` holds the masks of the training videos too).  Video class: second list
    field (SHT ``name,label``) or the ``abnormal`` / ``normal`` name prefix (UBnormal)."""
    device = next(enc.parameters()).device
    scores, labels, pool = [], [], []

    def flush():
        if not pool:
            return
        sc = scoring.ltn_sequence_scores(enc, head, [q for _, seqs, _ in pool for q in seqs]).cpu().numpy()
        off = 0
        for slot, seqs, expand in pool:
            scores[slot], labels[slot] = expand(sc[off:off + len(seqs)]); off += len(seqs)
        pool.clear()

    def frame_labels(abnormal, key, n):
        if not abnormal:
            return np.zeros(n)
        return np.asarray(np.load(mask_dir + key + ".npy", allow_pickle=True))[:n]

    with FeatureArchive(train_archive) as arc:
        for line in open(training_txt, "r").readlines():
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
            f = _dev(arc[key + ".npy"], device, n_patch)
            if mode != "LTN":
                sc = scoring.stn_clip_scores(enc, head, f).reshape(-1).cpu().numpy()
                s = np.repeat(sc, segment_len)
                scores.append(s); labels.append(frame_labels(abnormal, key, s.shape[0]))
                continue
            seqs, ranges = scoring.ltn_part_sequences(f, part_len, tail="rewindow")
            scores.append(None); labels.append(None)

            def expand(v, ranges=ranges, abnormal=abnormal, key=key):
                sfr = np.concatenate([np.full((e - b) * segment_len, float(x), np.float32) for x, (b, e) in zip(v, ranges)])
                return sfr, frame_labels(abnormal, key, sfr.shape[0])
            pool.append((len(scores) - 1, seqs, expand))
            if sum(len(q[1]) for q in pool) >= pool_sequences:
                flush()
        flush()
    s, l = np.concatenate(scores), np.concatenate(labels)
    auc = roc_auc(s, l)
    return (auc, s, l) if return_frames else auc
