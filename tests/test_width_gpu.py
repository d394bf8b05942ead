"""BASELINE configs 3 and 5 at the WIDTH they are quoted on, in the dtype they are quoted in (GPU only).

VERDICT r3 (row G3, What's missing #1, What's weak #2): the bf16 co-teaching chain and the mixed-batch AUC check ran at
d_model = 32 / 128 / 256, where ``functional.gemm``'s shape gate sends every product to the convert-while-staging kernel - the
production bf16 kernels (``gemm_bf16p_kernel`` on packed operands, ``attn_fwd3 / attn_bwd3`` of csrc/attention_pk.hip) never ran
in them.  Here every stage runs at d_model = 2048 (1024 for the UBnormal half), H = 8 x 256, F = 3027 / 4096, three layers, with
>= 256 sequences per step and token counts that fill whole 256-row pack tiles, and each test ASSERTS which kernels its launches
selected (a spy on ``functional._launch_gemm`` reads the descriptor's dtype, a spy on ``functional.attn_fwd`` sees packed Q|K|V).

* config 3: Train/spatio_transformer_shanghaitech.py -> pseudo_labels_generator_spatio.py -> temporal_transformer_shanghaitech.py
  -> pseudo_labels_generator_temporal.py -> spatio_transformer_MIL_CE.py (+ its end-of-round labels) ->
  Test/evaluation_shanghaitech_ubnormal.py, the entry points the command lines call (``lstc_vad_amd.cli``), on feature archives
  written at width; bf16 teacher-forced against the fp32 HIP chain (which the full-width golden tests pin step-wise to the
  reference: ``stn_full``, ``ltn_full_256``, ``stn_mil_ce_full``).  README.md:21-36 of the reference lists the command lines.
* config 5: ``engine.MixedStep`` over a UBnormal-shaped pair (d_model 1024, L = 5, S = 81) and an SHT-shaped pair (d_model 2048,
  L = 3, S = 49) - the same run in fp32 and bf16, AUC of held-out videos within 1e-2.
* inference with short tail parts (S = 33 / 17 under L = 3; 65 / 17 under L = 5) at width, fp32 AND bf16, against scores the
  reference itself produced (``eval_scores_*`` of the full-width fixtures; Train/pseudo_labels_generator_temporal.py:110-146).
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
SEG = 16
D, P = 2048, 16


# ------------------------------------------------------------------------------------------------ kernel-selection spies
class KernelSpy:
    """Records, while active, every lstc_gemm descriptor's (dtype, M, N, K, flops) and whether lstc_attn_fwd got packed operands -
    each tagged with whether it was launched from inside an optimisation step (``engine.TrainStep.step`` / ``MixedStep.step``:
    forward, loss, backward, Adagrad) or outside one (the in-loop evaluation of a Train script, the generators, the Test script)."""

    def __enter__(self):
        from lstc_vad_amd import engine
        from lstc_vad_amd import functional as Fn
        self.Fn, self.engine, self.gemms, self.attn, self.depth, self.steps = Fn, engine, [], [], 0, 0
        self._lg, self._af, self._ts, self._ms = Fn._launch_gemm, Fn.attn_fwd, engine.TrainStep.step, engine.MixedStep.step
        spy = self

        def launch(d, flops):
            spy.gemms.append((int(d.dtype), int(d.M), int(d.N), int(d.K), float(flops), spy.depth > 0))
            return spy._lg(d, flops)

        def attn_fwd(q, *a, **kw):
            spy.attn.append((isinstance(q, Fn.Packed), spy.depth > 0))
            return spy._af(q, *a, **kw)

        def in_step(real):
            def step(self_, *a, **kw):
                spy.depth += 1
                spy.steps += 1
                try:
                    return real(self_, *a, **kw)
                finally:
                    spy.depth -= 1
            return step
        Fn._launch_gemm, Fn.attn_fwd = launch, attn_fwd
        engine.TrainStep.step, engine.MixedStep.step = in_step(self._ts), in_step(self._ms)
        return self

    def __exit__(self, *exc):
        self.Fn._launch_gemm, self.Fn.attn_fwd = self._lg, self._af
        self.engine.TrainStep.step, self.engine.MixedStep.step = self._ts, self._ms
        return False

    def packed_flop_share(self, in_step=None):
        from lstc_vad_amd import _lib
        sel = [g for g in self.gemms if in_step is None or g[5] == in_step]
        tot = sum(g[4] for g in sel)
        return sum(g[4] for g in sel if g[0] == _lib.BF16P) / max(tot, 1.0)

    def assert_production_bf16(self, what, min_share=0.97):
        """The optimisation steps of ``what`` ran on the production bf16 kernels: >= ``min_share`` of their GEMM FLOPs on gemm_bf16p
        (the rest: heads and the CLS-only last layer, which stay on the small-product kernel by design) and EVERY attention core
        of a step on packed Q | K | V (csrc/attention_pk.hip)."""
        assert self.steps > 0 and self.packed_flop_share(True) >= min_share, (what, self.packed_flop_share(True), len(self.gemms))
        att = [pk for pk, st in self.attn if st]
        assert len(att) >= 2 * self.steps and all(att), (what, att)

    def assert_exact_f32(self, what):
        from lstc_vad_amd import _lib
        assert self.gemms and all(g[0] == _lib.F32 for g in self.gemms), what
        assert not any(pk for pk, _ in self.attn), what


# ------------------------------------------------------------------------------------------------ a feature world at width
def _wide_world(root):
    """SHT-dialect archive + lists + masks at d = 2048, P = 16: 8 normal + 8 abnormal training videos (one batch of 8 pairs;
    clip counts 49-60 with every residue mod 3, so the temporal generator meets tails of one and two clips) and 6 test videos.
    Abnormal videos carry a brighter stretch in their first d/8 channels; the masks mark it."""
    from lstc_vad_amd.archive import write_archive
    os.makedirs(root, exist_ok=True)
    g = torch.Generator().manual_seed(20240)
    train = [(f"{1 + i // 2:02d}_{'00' if i % 2 else '0'}{11 + i}", i % 2, 49 + (i * 5) % 12) for i in range(16)]
    test = [(f"{9 + i:02d}_{'00' if i % 2 else '0'}{31 + i}", i % 2, 40 + 7 * i + (i % 3)) for i in range(6)]
    arrays, W = {}, {"root": root}
    W["masks"] = os.path.join(root, "masks") + os.sep
    os.makedirs(W["masks"], exist_ok=True)
    for name, lab, n in train + test:
        f = 0.5 * torch.relu(torch.randn(n, P, D, generator=g))
        if lab:
            a, b = n // 3, n // 3 + max(n // 3, 2)
            f[a:b, :, : D // 8] += 0.3
            m = np.zeros(n * SEG + 5, np.float64)
            m[a * SEG:b * SEG] = 1.0
            np.save(os.path.join(W["masks"], name + ".npy"), m)
        arrays[name + ".npy"] = f.numpy()
    W["feats"] = write_archive(os.path.join(root, "feats.npz"), arrays)
    W["train_txt"] = os.path.join(root, "train.txt")
    open(W["train_txt"], "w").write("".join(f"{n},{l}\n" for n, l, _ in train))
    W["test_txt"] = os.path.join(root, "test.txt")
    open(W["test_txt"], "w").write("".join(f"{n},{l},{-1 if l else c * SEG + 3}\n" for n, l, c in test))
    W["train"], W["test"] = train, test
    return W


def _loss_rows(log_path, key, fields):
    rows = []
    for line in open(log_path).read().splitlines():
        if key in line and "]: " in line:
            body = line.split("]: ", 1)[1].replace(",", " ").split()
            vals = dict(zip(body[0::2], body[1::2]))
            rows.append([float(vals[f]) for f in fields])
    return np.array(rows)


def _gap_threshold(label_dict):
    """A pseudo-label threshold in the widest gap of the central half of the raw scores (upstream's 0.34 / 0.9 are tuned to
    trained models; after two optimisation steps every score sits near one value, and a threshold inside a dense stretch would
    measure which side of it rounding puts a score on, not the arithmetic)."""
    s = np.sort(np.unique(np.concatenate([np.asarray(v, np.float64).reshape(-1) for v in label_dict.values()])))
    lo, hi = int(0.25 * len(s)), max(int(0.75 * len(s)), int(0.25 * len(s)) + 2)
    mid = s[lo:hi]
    k = int(np.argmax(np.diff(mid)))
    return float(0.5 * (mid[k] + mid[k + 1])), float(mid[k + 1] - mid[k])


MODEL = ["--d_model", str(D), "--n_head", "8", "--d_k", "256", "--d_v", "256", "--n_patch", str(P)]
NO_DROP = ["--MHA_attn_dropout", "0", "--MHA_fc_dropout", "0", "--FFN_dropout", "0", "--position_dropout", "0"]
STEPS = 2
# Adagrad's first update moves EVERY weight by lr (lr * g / sqrt(g^2)): at d_model = 2048 the reference rates 1e-4 / 1e-2 shift a
# head pre-activation by ~1e-2 * 2048 * |x| ~ 4 in one step and the sigmoid saturates to exactly 0 / 1 on i.i.d. synthetic features
# (bench.py scales the rates for the same reason), and the three Linears of the head compound the coherent shift (3e-5
# still drove P(abnormal) to < 5e-5 after ONE step).  At 1e-6 / 1e-6 the scores stay in the open range of the sigmoid / softmax,
# so thresholds cut and differences show.
LR_ENC, LR_HEAD = "1e-6", "1e-6"


def _run_chain(W, out, dtype, feed=None, thresholds=None):
    """The five stages + the Test/ script through ``lstc_vad_amd.cli`` (what ``python Train/<script>.py <flags>`` calls), at
    width, ``STEPS`` optimisation steps per training stage, dropout off, in GEMM mode ``dtype``.  ``feed``: directory of another
    run whose checkpoints / label files every stage READS (teacher forcing); this run's own files go to ``out``."""
    from lstc_vad_amd import cli
    from lstc_vad_amd import functional as Fn
    os.makedirs(out, exist_ok=True)
    src = out if feed is None else feed
    j = lambda d, f: os.path.join(d, f)
    dt = ["--compute_dtype", dtype]
    data = ["--dataset", "SHT", "--dataset_path", W["feats"], "--training_txt", W["train_txt"], "--testing_txt", W["test_txt"],
            "--test_mask_dir", W["masks"], "--model_save_dir", j(out, "ck") + os.sep, "--seed", "5", "--inter_epoch", "1000",
            "--save_threshold", "2", "--batch_size", "8", "--part_num", "16", "--steps", str(STEPS)]
    res, spies = {"dir": out}, {}

    def stage(name, fn, *a):
        with KernelSpy() as spy:
            fn(*a)
            torch.cuda.synchronize()
        spies[name] = spy
        import gc
        gc.collect(); torch.cuda.empty_cache()
    try:
        # 1. STN, fresh xavier weights from --seed (identical in both modes)
        stage("stn", cli.train, "spatio_transformer_shanghaitech",
              MODEL + NO_DROP + data + dt + ["--epochs", "4", "--part_len", "4", "--n_hidden", "3027", "--FFN_layerNorm", "--train_dataset", W["feats"],
                                             "--encoder_weight_init", "--regressor_weight_init", "--regressor_dropout", "0",
                                             "--lr_encoder", LR_ENC, "--lr_regressor", LR_HEAD, "--save_final", j(out, "stn_"), "--log_dir", j(out, "l1")])
        res["stn"] = _loss_rows(j(out, "l1/spatio_transformer_shanghaitech.log"), "err", ["loss", "err", "l1"])
        # 2. its pseudo labels (raw scores first when the thresholds are still to be placed)
        gen = MODEL + ["--dataset", "SHT", "--dataset_path", W["feats"], "--training_txt", W["train_txt"], "--FFN_layerNorm"] + dt
        gen_s = gen + ["--n_hidden", "3027", "--spatio_model_path", j(src, "stn_encoder.ckpt"), "--regression_model_path", j(src, "stn_head.ckpt")]
        if thresholds is None:
            stage("raw_s", cli.generate_pseudo_labels, "pseudo_labels_generator_spatio", gen_s + ["--threshold", "-1", "--pseudo_labels_path", j(out, "raw_s.npy")])
            thr_s, gap_s = _gap_threshold(np.load(j(out, "raw_s.npy"), allow_pickle=True).tolist())
        else:
            thr_s, gap_s = thresholds[0], None
        stage("pl_s", cli.generate_pseudo_labels, "pseudo_labels_generator_spatio", gen_s + ["--threshold", repr(thr_s), "--pseudo_labels_path", j(out, "pl_s.npy")])
        res["pl_s"] = np.load(j(out, "pl_s.npy"), allow_pickle=True).tolist()
        # 3. LTN on those labels
        stage("ltn", cli.train, "temporal_transformer_shanghaitech",
              MODEL + NO_DROP + data + dt + ["--epochs", "4", "--part_len", "3", "--n_hidden", "4096", "--FFN_layerNorm", "--MHA_layerNorm",
                                             "--relative_position_encoding", "--encoder_weight_init", "--classifier_weight_init", "--classifier_dropout", "0",
                                             "--pseudo_labels_path", j(src, "pl_s.npy"), "--lr_encoder", LR_ENC, "--lr_classifier", LR_HEAD,
                                             "--save_final", j(out, "ltn_"), "--log_dir", j(out, "l3")])
        res["ltn"] = _loss_rows(j(out, "l3/temporal_transformer_shanghaitech.log"), "MIL_l1", ["CE_loss", "MIL_loss", "MIL_l1"])
        # 4. its pseudo labels: full parts S = 49, the videos' short tails as shorter sequences (S = 33, S = 17)
        gen_t = gen + ["--n_hidden", "4096", "--part_len", "3", "--MHA_layerNorm", "--relative_position_encoding",
                       "--temporal_model_path", j(src, "ltn_encoder.ckpt"), "--classifier_model_path", j(src, "ltn_head.ckpt")]
        if thresholds is None:
            stage("raw_t", cli.generate_pseudo_labels, "pseudo_labels_generator_temporal", gen_t + ["--threshold", "-1", "--pseudo_labels_path", j(out, "raw_t.npy")])
            thr_t, gap_t = _gap_threshold(np.load(j(out, "raw_t.npy"), allow_pickle=True).tolist())
        else:
            thr_t, gap_t = thresholds[1], None
        stage("pl_t", cli.generate_pseudo_labels, "pseudo_labels_generator_temporal", gen_t + ["--threshold", repr(thr_t), "--pseudo_labels_path", j(out, "pl_t.npy")])
        res["pl_t"] = np.load(j(out, "pl_t.npy"), allow_pickle=True).tolist()
        # 5. STN co-teaching (MIL + BCE on the LTN's labels) from the trained STN, then its end-of-round label file
        stage("mce", cli.train, "spatio_transformer_MIL_CE",
              MODEL + data + dt + ["--spatio_epochs", "4", "--spatio_part_len", "4", "--spatio_n_hidden", "3027", "--spatio_FFN_layerNorm", "--load_model",
                                   "--spatio_model_path", j(src, "stn_encoder.ckpt"), "--regression_model_path", j(src, "stn_head.ckpt"),
                                   "--spatio_pseudo_path", j(src, "pl_t.npy"), "--temporal_pseudo_path", j(out, "pl_mce"), "--threshold", repr(thr_s),
                                   "--spatio_MHA_attn_dropout", "0", "--spatio_MHA_fc_dropout", "0", "--spatio_FFN_dropout", "0", "--regressor_dropout", "0",
                                   "--lr_encoder", LR_ENC, "--lr_regressor", LR_HEAD, "--save_final", j(out, "mce_"), "--log_dir", j(out, "l5")])
        res["mce"] = _loss_rows(j(out, "l5/spatio_transformer_MIL_CE.log"), "spatio_loss", ["MIL_loss", "err", "l1", "CE_loss"])
        res["pl_mce"] = np.load(j(out, "pl_mce.npy"), allow_pickle=True).tolist()
        # 6. Test/evaluation_shanghaitech_ubnormal.py on the trained LTN (short tails re-windowed to full parts there)
        aucs = []
        stage("eval", lambda *a: aucs.append(cli.evaluate_cli(*a)), "evaluation_shanghaitech_ubnormal",
              ["--d_model", str(D), "--temporal_n_head", "8", "--temporal_d_k", "256", "--temporal_d_v", "256", "--temporal_n_hidden", "4096",
               "--temporal_MHA_layerNorm", "--temporal_FFN_layerNorm", "--temporal_relative_position_encoding", "--part_len", "3", "--dataset", "SHT",
               "--dataset_path", W["feats"], "--testing_txt", W["test_txt"], "--test_mask_dir", W["masks"],
               "--temporal_model_path", j(src, "ltn_encoder.ckpt"), "--classifier_model_path", j(src, "ltn_head.ckpt")] + dt)
        res["auc"] = float(aucs[0])
    finally:
        Fn.set_compute_dtype("fp32")
    res["thr"], res["gaps"], res["spies"] = (thr_s, thr_t), (gap_s, gap_t), spies
    return res


def test_coteaching_chain_at_production_width_bf16_runs_the_packed_kernels_and_tracks_fp32(tmp_path):
    """BASELINE config 3 ("Full STN -> pseudo-label -> LTN co-teaching loop, ShanghaiTech config, bf16") at d_model = 2048.
    Bars: every logged loss term of every step within 5e-2 of the fp32 chain's, label files with the same
    zero pattern on >= 98 % of their entries and values within 2e-2, LTN test AUC within 1e-2 - bf16 teacher-forced on the fp32
    chain's files; the training stages and the pooled inference launches must have run gemm_bf16p / attention_pk."""
    W = _wide_world(str(tmp_path / "world"))
    A = _run_chain(W, str(tmp_path / "fp32"), "fp32")
    for name in ("stn", "ltn", "mce"):
        A["spies"][name].assert_exact_f32(name)
    B = _run_chain(W, str(tmp_path / "bf16"), "bf16", feed=A["dir"], thresholds=A["thr"])
    # --- the production bf16 kernels ran: 1024 sequences x S = 17 (STN, MIL_CE), 256 x S = 49 (LTN) per step
    for name in ("stn", "ltn", "mce"):
        B["spies"][name].assert_production_bf16(name)
    # inference: the pooled launches are GEMM-wise on the packed kernel; their attention is packed where the pooled token count
    # fills whole 256-row tiles and exact otherwise (a shape property of the pool, not of the mode) - so only the GEMM share is held
    for name in ("pl_s", "pl_t", "eval"):
        assert B["spies"][name].steps == 0 and B["spies"][name].packed_flop_share() >= 0.9, (name, B["spies"][name].packed_flop_share())
    tails = {g[1] for g in B["spies"]["pl_t"].gemms}
    assert len(tails) >= 3, tails                          # full parts and both tail lengths went through as separate batches
    differs = 0.0
    for stage in ("stn", "ltn", "mce"):
        assert A[stage].shape == B[stage].shape and A[stage].shape[0] == STEPS and np.isfinite(B[stage]).all(), (stage, A[stage], B[stage])
        assert np.abs(np.diff(A[stage], axis=0)).max() > 0                               # the optimizer moved something between the steps
        dlt = np.abs(A[stage] - B[stage])
        assert dlt.max() < 5e-2, (stage, A[stage], B[stage])
        differs = max(differs, float(dlt.max()))
    for f in ("pl_s", "pl_t", "pl_mce"):
        assert list(A[f].keys()) == list(B[f].keys()) and len(A[f]) == 16
        same = total = 0
        worst = 0.0
        for k in A[f]:
            a, b = np.asarray(A[f][k], np.float32), np.asarray(B[f][k], np.float32)
            assert a.shape == b.shape and np.isfinite(b).all()
            same += int(((a > 0) == (b > 0)).sum()); total += a.size
            both = (a > 0) & (b > 0)
            if both.any():
                worst = max(worst, float(np.abs(a - b)[both].max()))
        assert worst < 2e-2, (f, worst)
        assert same >= 0.98 * total, (f, same, total, A["thr"], A["gaps"])
        assert 0 < sum(int((np.asarray(v) > 0).sum()) for v in A[f].values()) < total      # the threshold really cuts
        differs = max(differs, worst)
    assert abs(A["auc"] - B["auc"]) < 1e-2, (A["auc"], B["auc"])
    assert differs > 1e-6, "bf16 chain equals the fp32 chain to the last digit: the mode did not run"
    print(f"\n[config 3 at width] thresholds {A['thr']} (gaps {A['gaps']}), loss diff <= {differs:.2e}, AUC fp32 {A['auc']:.4f} bf16 {B['auc']:.4f}")


# ------------------------------------------------------------------------------------------------ config 5 at width
def test_mixed_step_at_production_width_bf16_vs_fp32_auc():
    """BASELINE config 5 ("UBnormal config (d_model=1024, part_len=5) mixed with SHT in one batch, bf16 + fp32 AUC parity
    check") at the production widths: engine.MixedStep over a UBnormal-shaped pair (d_model 1024, H = 8 x 256, L = 5: 256
    sequences x S = 81) and an SHT-shaped pair (d_model 2048, L = 3: 256 x S = 49), fused Q|K|V, 4 steps from identical weights
    and batches in fp32 and in bf16; both pairs then score 128 held-out sequences in the mode they were trained in.  Frame-level
    AUC per dataset within 1e-2 between the modes, scores within 5e-2, losses within 5e-2; the bf16 run on gemm_bf16p +
    attention_pk (asserted)."""
    from argparse import Namespace
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd.engine import MixedStep, TrainStep
    from lstc_vad_amd.metrics import roc_auc
    from lstc_vad_amd.models import Classifier, Encoder
    cfgs = [dict(d_model=1024, L=5), dict(d_model=2048, L=3)]
    bs, pn = 8, 16

    def run(mode):
        Fn.set_compute_dtype(mode)
        try:
            steps, batches, tests = [], [], []
            for ci, c in enumerate(cfgs):
                d, L = c["d_model"], c["L"]
                torch.manual_seed(100 + ci)
                enc = Encoder(n_layers=3, n_head=8, d_k=256, d_v=256, d_model=d, d_inner=4096, MHA_attn_dropout=0.0, MHA_fc_dropout=0.0,
                              FFN_dropout=0.0, MHA_layerNorm=True, FFN_layerNorm=True, relative_pe=True, window_size=4,
                              window_depth=L, weight_init=True).to(DEV).train()
                head = Classifier(d, 0.0).to(DEV).train()
                args = Namespace(batch_size=bs, part_num=pn, part_len=L, n_patch=P, lambda_1=0.01, lambda_MIL=1.0, lambda_CE=0.8,
                                 temporal_only=False, clip_grad=False)
                # rates: see LR_ENC / LR_HEAD above (the reference rates saturate the softmax on synthetic features within a step)
                steps.append(TrainStep(args, "LTN", enc, head, 1e-5, 3e-5, 1e-3, fuse_qkv="on"))      # (measured: AUC 0.96 / 0.89, unsaturated)
                g = torch.Generator(device=DEV).manual_seed(7 + ci)
                nf = 0.5 * torch.relu(torch.randn(bs, pn * L, P, d, device=DEV, generator=g))
                af = 0.5 * torch.relu(torch.randn(bs, pn * L, P, d, device=DEV, generator=g))
                af[:, :, :, : d // 8] += 0.05                                   # a faint anomaly signature: AUC well inside (0.5, 1)
                al = torch.ones(bs, pn * L, 1, device=DEV)
                batches.append((nf, af, al))
                xt = 0.5 * torch.relu(torch.randn(128, L * P, d, device=DEV, generator=g))
                xt[64:, :, : d // 8] += 0.05
                tests.append(xt)
            mixed = MixedStep(steps)
            with KernelSpy() as spy:
                for _ in range(4):
                    sc = mixed.step(batches)
                torch.cuda.synchronize()
            scores = []
            with torch.no_grad():
                for ts, xt in zip(steps, tests):
                    ts.encoder.eval(); ts.head.eval()
                    scores.append(ts.head(ts.encoder.forward_cls(xt))[:, 1].cpu().numpy())
            return [float(s[0]) for s in sc], scores, spy
        finally:
            Fn.set_compute_dtype("fp32")

    loss32, s32, spy32 = run("fp32")
    torch.cuda.empty_cache()
    loss16, s16, spy16 = run("bf16")
    spy32.assert_exact_f32("fp32 mixed step")
    spy16.assert_production_bf16("bf16 mixed step")
    labels = np.r_[np.zeros(64), np.ones(64)]
    for name, a, b, l32, l16 in zip(("UBnormal-shaped", "SHT-shaped"), s32, s16, loss32, loss16):
        assert np.isfinite(a).all() and np.isfinite(b).all()
        assert 1e-6 < np.max(np.abs(a - b)) < 5e-2, (name, np.max(np.abs(a - b)))
        assert a.std() > 1e-3                                            # unsaturated scores: the comparison can see something
        auc32, auc16 = roc_auc(a, labels), roc_auc(b, labels)
        assert abs(auc32 - auc16) < 1e-2, (name, auc32, auc16)
        assert abs(l32 - l16) < 5e-2, (name, l32, l16)
        print(f"\n[config 5 at width] {name}: AUC fp32 {auc32:.4f} bf16 {auc16:.4f}, score diff {np.max(np.abs(a - b)):.2e}, loss {l32:.4f} / {l16:.4f}")


# ------------------------------------------------------------------------------------------------ inference with short tails, at width
@pytest.mark.parametrize("name", ["ltn_full", "ltn_ubnormal_full", "ltn_ucf_full"])
@pytest.mark.parametrize("dtype,bar", [("fp32", 1e-4), ("f32x3", 1e-4), ("bf16", 2e-2)])
def test_short_tail_inference_at_width_matches_the_reference(name, dtype, bar):
    """The pseudo-label generator feeds a video's short tail as a SHORTER sequence (Train/pseudo_labels_generator_temporal.py:
    110-146): under L = 3 that is S = 33 and S = 17, under L = 5 S = 65 and 17, under UCF's L = 2 S = 10 - attention
    instantiations and bias-index slices ([:S-1, :S-1] of the [16 L, 16 L] index) that no training step uses.  Eval-mode
    P(abnormal) of 8 sequences at production width against what the reference itself computed (``eval_scores_*`` of the
    full-width fixtures): fp32 and f32x3 within the north_star tolerance 1e-4, bf16 within 2e-2; also through
    scoring.ltn_sequence_scores, the pooled path of pipeline.generate_pseudo_labels / evaluate_auc, with the three lengths mixed
    in one call."""
    from cases import FULL_CASES, fill_params
    from lstc_vad_amd import functional as Fn
    from lstc_vad_amd import scoring
    from lstc_vad_amd import synthetic as syn
    from lstc_vad_amd.models import Classifier, Encoder
    mode, ekw, skw, seed = FULL_CASES[name]
    z = np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)
    d, L, Pn = ekw["d_model"], skw["part_len"], skw["n_patch"]
    enc = Encoder(n_layers=3, MHA_attn_dropout=0.3, MHA_fc_dropout=0.3, FFN_dropout=0.3, position_dropout=0.3, weight_init=False, **ekw)
    head = Classifier(d, 0.6, weight_init=False)
    fill_params(enc, seed); fill_params(head, seed + 1)
    enc, head = enc.to(DEV).eval(), head.to(DEV).eval()
    nf, _, _, _ = syn.training_batch(skw["batch_size"], skw["part_num"], L, Pn, d, seed=seed, with_pseudo=True, threshold=0.6)
    x = torch.from_numpy(nf).view(skw["batch_size"] * skw["part_num"], L * Pn, d)[:8].to(DEV)
    views = {"full": x, "tail": x[:, :(L - 1) * Pn].contiguous(), "tail1": x[:, :Pn].contiguous()}
    Fn.set_compute_dtype(dtype)
    try:
        with torch.no_grad():
            worst = 0.0
            for tag, xs in views.items():
                got = head(enc.forward_cls(xs)).view(-1, 2).cpu().numpy()
                ref = z["eval_scores_" + tag].reshape(-1, 2)
                assert np.abs(got - ref).max() < bar, (tag, np.abs(got - ref).max())
                worst = max(worst, float(np.abs(got - ref).max()))
            seqs = [v[i] for i in range(8) for v in views.values()]               # lengths interleaved, as a pool of videos has them
            pooled = scoring.ltn_sequence_scores(enc, head, seqs).cpu().numpy().reshape(8, 3)
            for c, tag in enumerate(views):
                assert np.abs(pooled[:, c] - z["eval_scores_" + tag].reshape(-1, 2)[:, 1]).max() < bar, tag
            if dtype == "bf16":
                assert worst > 1e-6              # the bf16 products really ran
    finally:
        Fn.set_compute_dtype("fp32")
