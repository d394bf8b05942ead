"""File-level stages around the training step: pseudo-label generation and frame-level AUC evaluation on feature
archives, with the reference's file formats.

* ``generate_pseudo_labels``  - Train/pseudo_labels_generator_spatio.py:67-90 and
  Train/pseudo_labels_generator_temporal.py:66-146.  Output: ``{"<video>.npy": float32 [n, 1]}`` written with
  ``np.save`` (a pickled dict; read back with ``np.load(..., allow_pickle=True).tolist()``, utils/load_dataset.py:20),
  scores at or below ``threshold`` zeroed.  STN: one row per clip.  LTN (SHT / UBnormal): one row per clip, the part's
  score repeated.  LTN (UCF): 32 rows - one per bin.
* ``evaluate_auc``            - Test/evaluation_shanghaitech_ubnormal.py:69-96, Test/evaluation_UCF.py:47-88 and the
  in-loop evaluation of the train scripts: frame-level ROC-AUC over the whole test list.

Both stages keep features on the device and batch all parts of a video (``lstc_vad_amd.scoring``).
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
                           d_model=None, segment_len=16, classifier_head=False, out_path=None):
    """mode 'STN' | 'LTN'; dataset 'SHT' | 'UCF' | 'UBnormal'.  Returns the dict (and writes it when ``out_path``)."""
    device = next(enc.parameters()).device
    d_model = d_model or enc.layer_norm.normalized_shape[0]
    out = {}
    with FeatureArchive(dataset_path) as arc:
        for line, key in _train_keys(dataset, training_txt):
            if mode == "STN":
                s = scoring.stn_clip_scores(enc, head, _dev(arc[key + ".npy"], device), classifier_head).reshape(-1, 1)
            elif dataset == "UCF":
                feats, n_frames = UCF_train(line, dataset_path, segment_len)
                f = _dev(feats, device).view(-1, n_patch, d_model)
                sc, ranges, _ = scoring.ltn_ucf_bin_scores(enc, head, f, n_frames, part_len, segment_len,
                                                            normalize=False, rewindow=False)
                s = torch.cat([v.repeat(e - b) for v, (b, e) in zip(sc, ranges)]).reshape(-1, 1)
            else:
                sc, ranges = scoring.ltn_part_scores(enc, head, _dev(arc[key + ".npy"], device), part_len, tail="short")
                s = torch.cat([v.repeat(e - b) for v, (b, e) in zip(sc, ranges)]).reshape(-1, 1)
            s = torch.where(s > threshold, s, torch.zeros_like(s))
            out[key + ".npy"] = s.cpu().numpy()
    if out_path:
        np.save(out_path, out)
    return out


@torch.no_grad()
def evaluate_auc(enc, head, mode, dataset, dataset_path, testing_txt, masks, part_len, n_patch, segment_len=16,
                 return_frames=False):
    """``masks``: directory of ``<video>.npy`` frame masks (SHT / UBnormal, ``--test_mask_dir``) or the ground-truth
    archive (UCF, ``--test_mask_path``).  LTN only scores parts; STN scores clips (in-loop evaluation of the spatio
    scripts, Train/spatio_transformer_shanghaitech.py:118-150: every clip's score x segment_len)."""
    device = next(enc.parameters()).device
    scores, labels = [], []
    if dataset == "UCF":
        for line in open(testing_txt, "r").readlines():
            feats, anno, n_frames, _ = UCF_test(line, dataset_path, masks, segment_len, return_name=True)
            f = _dev(feats, device)
            f = f.view(-1, n_patch, f.shape[-1])
            if mode == "LTN":
                sc, ranges, r = scoring.ltn_ucf_bin_scores(enc, head, f, n_frames, part_len, segment_len,
                                                           normalize=True, rewindow=True)
                s, l = scoring.frame_scores_ucf(sc.cpu().numpy(), ranges, r, anno, segment_len)
            else:
                sc = scoring.stn_clip_scores(enc, head, f).reshape(-1).cpu().numpy()
                s = np.repeat(sc, segment_len)
                l = np.asarray(anno[:s.shape[0]])
            scores.append(s); labels.append(l)
    else:
        loader = shanghaitech_test if dataset in ("SHT", "MT_SHT") else UBnormal_test
        feats_l, _, annos = loader(testing_txt, masks, dataset_path)
        for feats, anno in zip(feats_l, annos):
            f = _dev(feats, device, n_patch)
            if mode == "LTN":
                sc, ranges = scoring.ltn_part_scores(enc, head, f, part_len, tail="rewindow")
                s, l = scoring.frame_scores_sht(sc.cpu().numpy(), ranges, anno, segment_len)
            else:
                sc = scoring.stn_clip_scores(enc, head, f).reshape(-1).cpu().numpy()
                s = np.repeat(sc, segment_len)
                l = np.asarray(anno[:s.shape[0]])
            scores.append(s); labels.append(l)
    s, l = np.concatenate(scores), np.concatenate(labels)
    auc = roc_auc(s, l)
    return (auc, s, l) if return_frames else auc
