"""Evaluation-stage throughput at full width (SURVEY.md 8f-2 measurement): frame-level scoring of a synthetic SHT-shaped
test set with (a) the batched recipe of lstc_vad_amd.scoring (all parts of a video per launch sequence, CLS-only last
layer) and (b) the reference's launch pattern (one part per launch sequence, full last layer), same model, same scores.
Prints one JSON line.  Run on an MI355X: python tools/eval_throughput.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstc_vad_amd import scoring                                     # noqa: E402
from lstc_vad_amd.models import Classifier, Encoder                  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    from lstc_vad_amd import functional as Fn0
    if len(sys.argv) > 1:
        Fn0.set_compute_dtype(sys.argv[1])          # fp32 (default) | f32x3 | bf16
    torch.manual_seed(0)
    enc = Encoder(n_layers=3, n_head=8, d_k=256, d_v=256, d_model=2048, d_inner=4096, MHA_layerNorm=True, FFN_layerNorm=True,
                  relative_pe=True, window_size=4, window_depth=3, weight_init=False).to(dev).eval()
    head = Classifier(2048).to(dev).eval()
    rs = np.random.RandomState(0)
    lengths = rs.randint(24, 161, size=64)                            # SHT test videos: 24..160 clips of 16 frames
    g = torch.Generator(device=dev).manual_seed(1)
    videos = [0.5 * torch.relu(torch.randn(int(n), 16, 2048, device=dev, generator=g)) for n in lengths]
    clips = int(lengths.sum())

    def batched():
        return [scoring.ltn_part_scores(enc, head, v, 3, "rewindow")[0] for v in videos]

    def pooled():
        seqs, counts = [], []
        for v in videos:
            q, _ = scoring.ltn_part_sequences(v, 3, "rewindow")
            seqs.extend(q); counts.append(len(q))
        sc = scoring.ltn_sequence_scores(enc, head, seqs, max_batch=2048)
        return list(torch.split(sc, counts))

    def one_by_one():
        out = []
        for v in videos:
            sc = []
            for b, e in scoring.part_ranges(v.shape[0], 3):
                part = v[e - 3:e] if e - b < 3 else v[b:e]
                sc.append(head(enc(part.reshape(1, -1, 2048))[:, 0, :]).view(-1, 2)[:, 1])
            out.append(torch.cat(sc))
        return out

    res = {}
    with torch.no_grad():
        for name, fn in (("pooled", pooled), ("batched", batched), ("one_part_per_launch", one_by_one)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0, out)
    diff = max(float((a - b).abs().max()) for a, b in zip(res["batched"][1], res["one_part_per_launch"][1]))
    # data feed: one headline batch (2 x 32 videos x 96 clips x 16 x 2048 f32 = 805 MB) gathered out of an HBM-resident bank
    from lstc_vad_amd import functional as Fn
    bank = torch.cat(videos)                                              # [6061 clips, 16, 2048]
    idx = torch.from_numpy(rs.randint(0, bank.shape[0], size=2 * 32 * 96)).to(dev)
    out = Fn.gather_rows(bank, idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        Fn.gather_rows(bank, idx, out=out)
    e1.record(); torch.cuda.synchronize()
    gms = e0.elapsed_time(e1) / 10
    gather = {"batch_MB": round(out.numel() * 4 / 1e6, 1), "ms": round(gms, 4),
              "GBps_read_plus_write": round(2 * out.numel() * 4 / (gms * 1e-3) / 1e9, 1), "hbm_peak_GBps": 8000}
    print(json.dumps({"gemm_mode": Fn0.get_compute_dtype(), "workload": "LTN-SHT frame-level scoring, 64 synthetic test videos, %d clips, d=2048, S=49" % clips,
                      "pooled_across_videos_clips_per_s": round(clips / res["pooled"][0], 1),
                      "pooled_max_abs_diff_vs_per_video": max(float((a - b).abs().max()) for a, b in zip(res["pooled"][1], res["batched"][1])),
                      "batched_per_video_clips_per_s": round(clips / res["batched"][0], 1),
                      "reference_launch_pattern_clips_per_s": round(clips / res["one_part_per_launch"][0], 1),
                      "speedup_pooled_vs_reference_pattern": round(res["one_part_per_launch"][0] / res["pooled"][0], 2),
                      "max_abs_score_diff": diff, "lstc_gather_rows": gather}))


if __name__ == "__main__":
    main()
