"""Video scoring recipes of the reference's evaluation loops and pseudo-label generators, batched for the device.

The reference pushes ONE part of one video through the model per launch sequence (batch 1, e.g.
Test/evaluation_shanghaitech_ubnormal.py:75-92).  Sequences are independent, so here every part of a video that has
the same length goes through the encoder as one batch - and ``pipeline`` pools the parts of many videos into those
batches (``ltn_sequence_scores``; ``Encoder.forward_cls``: only the CLS row is computed in the last layer); results
are identical row by row (``test_full_width_scores_match_oracle`` checks batch invariance).

Recipes (all return per-part scores plus the ``(beg, end)`` clip ranges they stand for):

* ``stn_clip_scores``      - STN: one score per clip (Train/pseudo_labels_generator_spatio.py:79-86);
* ``ltn_part_scores``      - LTN on consecutive parts of ``part_len`` clips; the short tail part is either fed as a
                             shorter sequence (``tail='short'``, Train/pseudo_labels_generator_temporal.py:120-141) or
                             re-windowed to the video's last ``part_len`` clips (``tail='rewindow'``,
                             Test/evaluation_shanghaitech_ubnormal.py:83-84 and the in-loop evaluation
                             Train/temporal_transformer_shanghaitech.py:176-179) - including the reference's behaviour
                             for videos shorter than one part (a negative slice start);
* ``ltn_ucf_bin_scores``   - UCF: clips averaged into 32 bins (``np.linspace(0, n_frames // segment_len, 33)``), parts
                             of ``part_len`` bins, optional L2 normalisation (Test/evaluation_UCF.py:54-77 normalises,
                             Train/pseudo_labels_generator_temporal.py:73-99 does not).
"""
from __future__ import annotations

import numpy as np
import torch


def part_ranges(n: int, part_len: int):
    """[(beg, end)] of consecutive parts; the last one ends at ``n`` and may be shorter."""
    n_parts = n // part_len + (1 if (n // part_len) * part_len < n else 0)
    return [(i * part_len, n if i == n_parts - 1 else (i + 1) * part_len) for i in range(n_parts)]


def ltn_sequence_scores(enc, head, seqs, max_batch=4096):
    """seqs: list of [len_i * P, d] tensors (any mix of videos) -> tensor [len(seqs)] of P(abnormal).  Same-length
    sequences share a batch of up to ``max_batch`` rows, so a whole test set is scored in a handful of launch sequences
    that fill the GPU like a training step does."""
    out = torch.empty(len(seqs), device=seqs[0].device, dtype=torch.float32)
    by_len = {}
    for i, s in enumerate(seqs):
        by_len.setdefault(s.shape[0], []).append(i)
    for ids in by_len.values():
        for c in range(0, len(ids), max_batch):
            chunk = ids[c:c + max_batch]
            x = torch.stack([seqs[i] for i in chunk])
            out[torch.tensor(chunk, device=out.device)] = head(enc.forward_cls(x)).view(-1, 2)[:, 1]
    return out


_ltn_scores = ltn_sequence_scores


def stn_clip_scores(enc, head, feats, classifier_head=False):
    """feats [n_clips, P, d] -> [n_clips, 1].  ``classifier_head``: the generator's ``n_layers == 1`` branch pairs the
    encoder with a Classifier and keeps column 1 (Train/pseudo_labels_generator_spatio.py:81-82)."""
    y = head(enc.forward_cls(feats))
    return y[:, 1] if classifier_head else y


def ltn_part_sequences(feats, part_len, tail="rewindow"):
    """feats [n_clips, P, d] -> (list of [len*P, d] sequences, [(beg, end)])."""
    n, P, d = feats.shape
    ranges = part_ranges(n, part_len)
    seqs = []
    for beg, end in ranges:
        if end - beg < part_len and tail == "rewindow":
            part = feats[end - part_len:end]              # python slice semantics, as upstream (negative start allowed)
        else:
            part = feats[beg:end]
        seqs.append(part.reshape(-1, d))
    return seqs, ranges


def ltn_part_scores(enc, head, feats, part_len, tail="rewindow"):
    """feats [n_clips, P, d] -> (scores [n_parts], [(beg, end)])."""
    seqs, ranges = ltn_part_sequences(feats, part_len, tail)
    return ltn_sequence_scores(enc, head, seqs), ranges


def ucf_bins(feats, n_frames, segment_len=16, max_clips=32):
    """feats [n_clips, P, d] -> (bins [32, P, d], r [33]): bin i is the mean of clips r[i]:r[i+1] (or clip r[i] when the
    range is empty)."""
    n_clips = n_frames // segment_len
    r = np.linspace(0, n_clips, max_clips + 1, dtype=np.int32)
    rows = []
    for i in range(max_clips):
        rows.append(feats[int(r[i])] if r[i] == r[i + 1] else feats[int(r[i]):int(r[i + 1])].mean(dim=0))
    return torch.stack(rows), r


def ucf_bin_ranges(part_len, rewindow=True, max_clips=32):
    """[(beg, end)] in bin units of the parts a UCF video is scored in - a function of the flags alone (every video has 32
    bins), so a rank that does not own a video still knows how many scores it contributes (pipeline, sharded passes)."""
    ranges = []
    for beg, end in part_ranges(max_clips, part_len):
        if rewindow and end - beg < part_len:
            beg = end - part_len
        ranges.append((beg, end))
    return ranges


def ucf_bin_edges(n_frames, segment_len=16, max_clips=32):
    """r [33]: bin i covers clips r[i]:r[i+1] (Test/evaluation_UCF.py:54)."""
    return np.linspace(0, n_frames // segment_len, max_clips + 1, dtype=np.int32)


def ltn_ucf_bin_sequences(feats, n_frames, part_len, segment_len=16, normalize=True, rewindow=True, max_clips=32):
    """-> (sequences, [(beg, end)] in bin units, r).  ``rewindow`` moves a short tail part back so that it spans
    ``part_len`` bins (Test/evaluation_UCF.py:66-67 - there ``beg`` itself moves, so the frames it labels move too)."""
    bins, r = ucf_bins(feats, n_frames, segment_len, max_clips)
    d = bins.shape[-1]
    ranges = ucf_bin_ranges(part_len, rewindow, max_clips)
    seqs = [bins[b:e].reshape(-1, d) for b, e in ranges]
    if normalize:
        seqs = [torch.nn.functional.normalize(s, p=2, dim=-1) for s in seqs]
    return seqs, ranges, r


def ltn_ucf_bin_scores(enc, head, feats, n_frames, part_len, segment_len=16, normalize=True, rewindow=True, max_clips=32):
    seqs, ranges, r = ltn_ucf_bin_sequences(feats, n_frames, part_len, segment_len, normalize, rewindow, max_clips)
    return ltn_sequence_scores(enc, head, seqs), ranges, r


def frame_scores_sht(scores, ranges, anno, segment_len=16):
    """Expand per-part scores to frame level and cut the matching labels (…ubnormal.py:89-91)."""
    s = np.concatenate([np.full((e - b) * segment_len, float(v), np.float32) for v, (b, e) in zip(scores, ranges)])
    n = ranges[-1][1] * segment_len
    return s, np.asarray(anno[:n])


def frame_scores_ucf(scores, ranges, r, anno, segment_len=16):
    """Test/evaluation_UCF.py:82-85: part (beg, end) covers frames r[beg]*seg : r[end]*seg."""
    s, l = [], []
    for v, (b, e) in zip(scores, ranges):
        s.append(np.full(int(r[e] - r[b]) * segment_len, float(v), np.float32))
        l.append(np.asarray(anno[int(r[b]) * segment_len:int(r[e]) * segment_len]))
    return np.concatenate(s), np.concatenate(l)
