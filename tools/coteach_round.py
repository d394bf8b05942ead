#!/usr/bin/env python3
"""BASELINE config 3 timed: ONE round of the co-teaching loop of the reference's README.md:21-36 at the ShanghaiTech width

    spatio_transformer_shanghaitech  ->  pseudo_labels_generator_spatio  ->  temporal_transformer_shanghaitech
        ->  pseudo_labels_generator_temporal  ->  spatio_transformer_MIL_CE  ->  Test/evaluation_shanghaitech_ubnormal

through the entry points the command lines call (``lstc_vad_amd.cli``), chained by the files upstream chains them by, on a
synthetic SHT-dialect world written to a scratch directory (feature archive, list files, frame masks).  Sizes: the reference's
own defaults - ``--batch_size 40`` pairs, ``--part_num 16``, STN ``--part_len 7`` (8960 sequences of S = 17 per step), LTN
``--part_len 3`` (1280 sequences of S = 49), d_model 2048, H = 8 x 256, n_hidden 3027 / 4096 - and ``steps`` optimisation steps
per training stage.  Reports steady-state steps/s and snippets/s of the three training stages (from the end of the first step
on: ``cli.LAST_RUN``), clips/s of the two generators and of the test-set evaluation, and the wall seconds of the round
(process start-up, model construction and archive reads included).  ``bench.py`` embeds the result as ``coteach_loop``.

    python tools/coteach_round.py [--dtype bf16] [--steps 6] [--pairs 48] [--out DIR]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEG, D, P = 16, 2048, 16


def build_world(root, pairs, n_test, seed=7):
    """``pairs`` normal + ``pairs`` abnormal training videos of 48-80 clips, ``n_test`` test videos; abnormal videos carry a
    brighter stretch in their first d/8 channels (marked by their frame masks)."""
    import numpy as np
    import torch
    from lstc_vad_amd.archive import write_archive
    os.makedirs(root, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    rs = np.random.RandomState(seed)
    train = [(f"{1 + i // 2:02d}_{'00' if i % 2 else '0'}{101 + i}", i % 2, int(rs.randint(48, 81))) for i in range(2 * pairs)]
    test = [(f"{90 + i:02d}_{'00' if i % 2 else '0'}{501 + i}", i % 2, int(rs.randint(48, 81))) for i in range(n_test)]
    W = {"root": root, "masks": os.path.join(root, "masks") + os.sep}
    os.makedirs(W["masks"], exist_ok=True)
    arrays = {}
    for name, lab, n in train + test:
        f = 0.5 * torch.relu(torch.randn(n, P, D, generator=g))
        if lab:
            a, b = n // 3, n // 3 + max(n // 3, 2)
            f[a:b, :, : D // 8] += 0.3
            m = np.zeros(n * SEG + 5, np.float64)
            m[a * SEG:b * SEG] = 1.0
            np.save(os.path.join(W["masks"], name + ".npy"), m)
        arrays[name + ".npy"] = f.numpy()
    W["feats"] = write_archive(os.path.join(root, "feats"), arrays)        # a directory archive: one .npy per video, memory-mapped
    W["train_txt"] = os.path.join(root, "train.txt")
    open(W["train_txt"], "w").write("".join(f"{n},{l}\n" for n, l, _ in train))
    W["test_txt"] = os.path.join(root, "test.txt")
    open(W["test_txt"], "w").write("".join(f"{n},{l},{-1 if l else c * SEG + 3}\n" for n, l, c in test))
    W["train_clips"] = sum(c for _, _, c in train)
    W["test_clips"] = sum(c for _, _, c in test)
    return W


def run_round(out, dtype="bf16", steps=6, pairs=48, n_test=16, batch_size=40, lr_scale=1e-2):
    """-> dict (see the module docstring).  ``lr_scale`` multiplies the reference learning rates: on i.i.d. synthetic features
    the reference rates saturate the heads within two Adagrad steps; timing does not depend on it."""
    import torch
    from lstc_vad_amd import cli
    from lstc_vad_amd import functional as Fn
    t_round = time.perf_counter()
    W = build_world(os.path.join(out, "world"), pairs, n_test)
    t_world = time.perf_counter() - t_round
    j = lambda f: os.path.join(out, f)
    model = ["--d_model", str(D), "--n_head", "8", "--d_k", "256", "--d_v", "256", "--n_patch", str(P)]
    dt = ["--compute_dtype", dtype]
    data = ["--dataset", "SHT", "--dataset_path", W["feats"], "--training_txt", W["train_txt"], "--testing_txt", W["test_txt"],
            "--test_mask_dir", W["masks"], "--model_save_dir", j("ck") + os.sep, "--seed", "5", "--inter_epoch", "100000",
            "--save_threshold", "2", "--batch_size", str(batch_size), "--part_num", "16", "--steps", str(steps)]
    lr_e, lr_h = repr(1e-4 * lr_scale), repr(1e-2 * lr_scale)
    res = {"dtype": dtype, "act_dtype": ("bf16" if Fn._ACT16 else "fp32") if dtype == "bf16" else None,
           "sizes": {"train_videos": 2 * pairs, "train_clips": W["train_clips"], "test_videos": n_test, "test_clips": W["test_clips"],
                     "batch_size_pairs": batch_size, "part_num": 16, "steps_per_training_stage": steps, "d_model": D,
                     "stn": "part_len 7, S = 17, n_hidden 3027: %d sequences per step" % (2 * batch_size * 16 * 7),
                     "ltn": "part_len 3, S = 49, n_hidden 4096, relative position bias: %d sequences per step" % (2 * batch_size * 16)},
           "world_build_s": round(t_world, 2)}

    def stage(name, fn, *a):
        t0 = time.perf_counter()
        r = fn(*a)
        torch.cuda.synchronize()
        rec = dict(cli.LAST_RUN)
        rec["wall_s"] = round(time.perf_counter() - t0, 3)
        if "steady_s" in rec and rec["steady_steps"] > 0:
            rec["steps_per_s"] = round(rec["steady_steps"] / rec["steady_s"], 2)
            rec["ms_per_step"] = round(1e3 * rec["steady_s"] / rec["steady_steps"], 3)
            rec["snippets_per_s"] = round(rec["snippets_per_step"] * rec["steady_steps"] / rec["steady_s"], 1)
            rec["steady_s"] = round(rec["steady_s"], 4)
        if rec.get("clips") and rec.get("score_s"):
            rec["clips_per_s"] = round(rec["clips"] / rec["score_s"], 1)
            rec["score_s"] = round(rec["score_s"], 4)
        rec.pop("script", None)
        res[name] = rec
        import gc
        gc.collect(); torch.cuda.empty_cache()
        return r
    try:
        stage("stn_train", cli.train, "spatio_transformer_shanghaitech",
              model + data + dt + ["--epochs", str(steps), "--part_len", "7", "--n_hidden", "3027", "--FFN_layerNorm", "--train_dataset", W["feats"],
                                   "--encoder_weight_init", "--regressor_weight_init", "--lr_encoder", lr_e, "--lr_regressor", lr_h,
                                   "--save_final", j("stn_"), "--log_dir", j("l1")])
        gen = model + ["--dataset", "SHT", "--dataset_path", W["feats"], "--training_txt", W["train_txt"], "--FFN_layerNorm"] + dt
        stage("stn_labels", cli.generate_pseudo_labels, "pseudo_labels_generator_spatio",
              gen + ["--n_hidden", "3027", "--spatio_model_path", j("stn_encoder.ckpt"), "--regression_model_path", j("stn_head.ckpt"),
                     "--threshold", "0.5", "--pseudo_labels_path", j("pl_s.npy")])
        stage("ltn_train", cli.train, "temporal_transformer_shanghaitech",
              model + data + dt + ["--epochs", str(steps), "--part_len", "3", "--n_hidden", "4096", "--FFN_layerNorm", "--MHA_layerNorm",
                                   "--relative_position_encoding", "--encoder_weight_init", "--classifier_weight_init",
                                   "--pseudo_labels_path", j("pl_s.npy"), "--lr_encoder", lr_e, "--lr_classifier", lr_h,
                                   "--save_final", j("ltn_"), "--log_dir", j("l3")])
        stage("ltn_labels", cli.generate_pseudo_labels, "pseudo_labels_generator_temporal",
              gen + ["--n_hidden", "4096", "--part_len", "3", "--MHA_layerNorm", "--relative_position_encoding",
                     "--temporal_model_path", j("ltn_encoder.ckpt"), "--classifier_model_path", j("ltn_head.ckpt"),
                     "--threshold", "0.5", "--pseudo_labels_path", j("pl_t.npy")])
        stage("stn_mil_ce_train", cli.train, "spatio_transformer_MIL_CE",
              model + data + dt + ["--spatio_epochs", str(steps), "--spatio_part_len", "7", "--spatio_n_hidden", "3027", "--spatio_FFN_layerNorm",
                                   "--load_model", "--spatio_model_path", j("stn_encoder.ckpt"), "--regression_model_path", j("stn_head.ckpt"),
                                   "--spatio_pseudo_path", j("pl_t.npy"), "--temporal_pseudo_path", j("pl_mce"), "--threshold", "0.5",
                                   "--lr_encoder", lr_e, "--lr_regressor", lr_h, "--save_final", j("mce_"), "--log_dir", j("l5")])
        aucs = []
        stage("ltn_test_eval", lambda *a: aucs.append(cli.evaluate_cli(*a)), "evaluation_shanghaitech_ubnormal",
              ["--d_model", str(D), "--temporal_n_head", "8", "--temporal_d_k", "256", "--temporal_d_v", "256", "--temporal_n_hidden", "4096",
               "--temporal_MHA_layerNorm", "--temporal_FFN_layerNorm", "--temporal_relative_position_encoding", "--part_len", "3",
               "--dataset", "SHT", "--dataset_path", W["feats"], "--testing_txt", W["test_txt"], "--test_mask_dir", W["masks"],
               "--temporal_model_path", j("ltn_encoder.ckpt"), "--classifier_model_path", j("ltn_head.ckpt")] + dt)
        res["ltn_test_eval"]["auc"] = float(aucs[0])
    finally:
        Fn.set_compute_dtype("fp32")
    res["round_wall_s"] = round(time.perf_counter() - t_round, 2)
    res["note"] = ("one co-teaching round (README.md:21-36) through the lstc_vad_amd.cli entry points, stages chained by their files; "
                   "steps_per_s / snippets_per_s = steady state from the end of a stage's first step; clips_per_s = scoring of the whole "
                   "list between device synchronisations; round_wall_s includes world construction, model construction, archive reads, "
                   "checkpoint and label files")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16", "f32x3"])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--pairs", type=int, default=48)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    out = a.out or tempfile.mkdtemp(prefix="lstc_coteach_")
    real = os.dup(1)
    os.dup2(2, 1)                      # the CLI entry points print to stdout: keep the JSON line alone there
    try:
        r = run_round(out, a.dtype, a.steps, a.pairs)
    finally:
        if not a.out:
            shutil.rmtree(out, ignore_errors=True)
    os.write(real, (json.dumps(r) + "\n").encode())


if __name__ == "__main__":
    main()
