"""GPU parity of the stages around the training step: the HIP model behind pseudo-label generation,
evaluation and the HBM-resident batch source, against fixtures produced by the reference's own scripts
(tests/golden/make_golden_pipeline.py) and against the host dataset path."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import pipeline_world as pw                                   # noqa: E402
from pipeline_cases import CHAIN, DATASET_CASES, build_dataset, chain_argv, weight_fingerprint   # noqa: E402
from lstc_vad_amd import functional as F                      # noqa: E402
from lstc_vad_amd import load_dataset as ds_mod               # noqa: E402
from lstc_vad_amd import pipeline                             # noqa: E402
from lstc_vad_amd.models import Classifier, Encoder, Regressor  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(HERE, "golden", "pipeline.npz"))
DEV = torch.device("cuda", 0)
TOL = 2e-5          # fp32 scores through 3 layers; the reference side ran torch-CPU


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    return pw.build(str(tmp_path_factory.mktemp("world")), Encoder, Regressor, Classifier)


def _load(world, enc_ckpt, enc_kw, head_ckpt, head_cls):
    strip = lambda sd: {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    enc = Encoder(**enc_kw)
    enc.load_state_dict(strip(torch.load(world[enc_ckpt], map_location="cpu")), strict=True)
    head = head_cls(32)
    head.load_state_dict(strip(torch.load(world[head_ckpt], map_location="cpu")), strict=True)
    return enc.to(DEV).eval(), head.to(DEV).eval()


def _check_pseudo(out, prefix, thr):
    keys = [k[len(prefix):] for k in G.files if k.startswith(prefix)]
    assert list(out.keys()) == keys
    for k in keys:
        ref, got = G[prefix + k], out[k]
        assert got.shape == ref.shape and got.dtype == np.float32
        near = np.abs(np.where(ref > 0, ref, got) - thr) < 1e-4       # scores sitting on the threshold may flip
        assert np.all(np.abs(got - ref)[~near] < TOL), (k, np.abs(got - ref).max())


def test_pseudo_labels_match_reference_generators(world, tmp_path):
    enc, head = _load(world, "ltn_sht_enc.ckpt", pw.LTN_SHT, "ltn_sht_cls.ckpt", Classifier)
    p = str(tmp_path / "pl.npy")
    out = pipeline.generate_pseudo_labels(enc, head, "LTN", "SHT", world["sht_feats"], world["sht_train"], 0.45, part_len=3,
                                          out_path=p)
    _check_pseudo(out, "pl/t_sht/", 0.45)
    back = np.load(p, allow_pickle=True).tolist()
    assert list(back) == list(out)
    out = pipeline.generate_pseudo_labels(enc, head, "LTN", "UBnormal", world["ubn_feats"], world["ubn_train"], 0.45, part_len=3)
    _check_pseudo(out, "pl/t_ubn/", 0.45)
    enc, head = _load(world, "ltn_ucf_enc.ckpt", pw.LTN_UCF, "ltn_ucf_cls.ckpt", Classifier)
    out = pipeline.generate_pseudo_labels(enc, head, "LTN", "UCF", world["ucf_feats"], world["ucf_train"], 0.25, part_len=2,
                                          n_patch=9)
    _check_pseudo(out, "pl/t_ucf/", 0.25)
    enc, head = _load(world, "stn_sht_enc.ckpt", pw.STN_SHT, "stn_sht_reg.ckpt", Regressor)
    out = pipeline.generate_pseudo_labels(enc, head, "STN", "SHT", world["sht_feats"], world["sht_train"], 0.34)
    _check_pseudo(out, "pl/s_sht/", 0.34)


def test_evaluation_matches_reference_scripts(world):
    enc, head = _load(world, "ltn_sht_enc.ckpt", pw.LTN_SHT, "ltn_sht_cls.ckpt", Classifier)
    for tag, dataset, txt, masks, feats in (("sht", "SHT", "sht_test", "sht_masks", "sht_feats"),
                                             ("ubn", "UBnormal", "ubn_test", "ubn_masks", "ubn_feats")):
        auc, s, l = pipeline.evaluate_auc(enc, head, "LTN", dataset, world[feats], world[txt], world[masks], 3, 16,
                                          return_frames=True)
        assert np.max(np.abs(s - G[f"ev/{tag}/scores"])) < TOL
        assert np.array_equal(l, G[f"ev/{tag}/labels"])
        assert abs(auc - float(G[f"ev/{tag}/auc"][0])) < 1e-6
    enc, head = _load(world, "ltn_ucf_enc.ckpt", pw.LTN_UCF, "ltn_ucf_cls.ckpt", Classifier)
    auc, s, l = pipeline.evaluate_auc(enc, head, "LTN", "UCF", world["ucf_feats"], world["ucf_test"], world["ucf_gt"], 2, 9,
                                      return_frames=True)
    assert np.max(np.abs(s - G["ev/ucf/scores"])) < TOL and np.array_equal(l, G["ev/ucf/labels"])
    assert abs(auc - float(G["ev/ucf/auc"][0])) < 1e-6


def test_gather_rows_is_an_exact_row_copy():
    g = torch.Generator().manual_seed(3)
    for rows, shape in ((37, (4,)), (200, (16, 32)), (64, (16, 2048))):
        bank = torch.randn((rows,) + shape, generator=g).to(DEV)
        idx = torch.randint(0, rows, (3 * rows + 5,), generator=g).to(DEV)
        assert torch.equal(F.gather_rows(bank, idx), bank[idx])
    with pytest.raises(RuntimeError):
        F.gather_rows(torch.zeros(4, 3, device=DEV), torch.zeros(2, dtype=torch.int64, device=DEV))     # 3 floats: not x4


@pytest.mark.parametrize("name", ["sh_uniform", "sh_random_pseudo", "ubn_uniform", "ubn_random", "sh_mutual_uniform_pseudo", "sh_mutual_random",
                                  "ucf_uniform", "ucf_random_pseudo", "sh_tencrop_uniform", "sh_tencrop_random", "ubn_tencrop", "ucf_crop_return",
                                  "sh_npatch1"])
def test_resident_pairs_serve_the_host_batches(world, name):
    """Batches gathered out of HBM == default-collated dataset items with the same seeds (which the CPU suite pins to the
    reference's classes) - round 5: also for the LAZY single-crop datasets (the co-teaching stage's MutualTraining class and UCF,
    whose items upstream re-reads from the archive one by one); round 6: the TEN-CROP classes (utils/load_dataset.py:134-232,
    :631-729) and UCF's ``crop_return`` (:437-438): all ten crops resident, the per-item / per-video crop draw part of the row index."""
    spec = DATASET_CASES[name]
    np.random.seed(spec["seed"]); random.seed(spec["seed"])
    host = build_dataset(ds_mod, spec, world)
    bs = 2
    host_batches = []
    for _ in range(2):
        for b in range(len(host) // bs):
            items = [host[b * bs + j] for j in range(bs)]
            host_batches.append([torch.stack([it[k].reshape(it[k].shape[0], -1) if k % 2 else it[k] for it in items])
                                 for k in range(4)])
        host.shuffle_keys()
    np.random.seed(spec["seed"]); random.seed(spec["seed"])
    res = ds_mod.ResidentPairs(build_dataset(ds_mod, spec, world), bs, DEV)
    got = []
    for _ in range(2):
        got.extend([[t.cpu() for t in batch] for batch in res])
        res.shuffle_keys()
    assert len(got) == len(host_batches) > 0
    for a, b in zip(got, host_batches):
        for k in range(4):
            assert torch.equal(a[k], b[k]), (name, k)


def _run(script_dir, script, argv):
    env = dict(os.environ, PYTHONPATH=ROOT, HIP_VISIBLE_DEVICES="0")
    return subprocess.run([sys.executable, script] + argv, cwd=os.path.join(ROOT, script_dir), env=env, capture_output=True,
                          text=True, timeout=600)


def test_reference_command_lines_on_feature_archives(world, tmp_path):
    """The Train/ and Test/ entry points with the reference's own flags, on archive files: generator -> LTN training with
    those pseudo labels -> evaluation script on the saved checkpoint."""
    model = ["--d_model", "32", "--n_head", "2", "--d_k", "16", "--d_v", "16", "--n_hidden", "64", "--MHA_layerNorm",
             "--FFN_layerNorm", "--relative_position_encoding", "--part_len", "3"]
    pl = str(tmp_path / "pl_cli.npy")
    r = _run("Train", "pseudo_labels_generator_temporal.py", model + [
        "--dataset", "SHT", "--dataset_path", world["sht_feats"], "--training_txt", world["sht_train"], "--data_parallel",
        "--temporal_model_path", world["ltn_sht_enc.ckpt"], "--classifier_model_path", world["ltn_sht_cls.ckpt"],
        "--pseudo_labels_path", pl, "--threshold", "0.45"])
    assert r.returncode == 0, r.stderr[-2000:]
    _check_pseudo(np.load(pl, allow_pickle=True).tolist(), "pl/t_sht/", 0.45)
    save = str(tmp_path / "ckpt") + os.sep            # upstream concatenates strings: the directory flag ends with a separator
    r = _run("Train", "temporal_transformer_shanghaitech.py", model + [
        "--dataset_path", world["sht_feats"], "--training_txt", world["sht_train"], "--testing_txt", world["sht_test"],
        "--test_mask_dir", world["sht_masks"], "--pseudo_labels_path", pl, "--batch_size", "2", "--part_num", "3",
        "--epochs", "2", "--inter_epoch", "1", "--model_save_dir", save, "--saved_prefix", "", "--save_threshold", "0",
        "--log_dir", str(tmp_path / "log")])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "best_test_AUC" in r.stderr and "best_train_AUC" in r.stderr and "MIL_loss" in r.stderr
    train_log = r.stderr
    saved = sorted(os.listdir(save))
    enc_ckpt = [f for f in saved if f.startswith("temporal_model")][-1]
    cls_ckpt = [f for f in saved if f.startswith("classifier_model")][-1]
    ev = ["--d_model", "32", "--temporal_n_head", "2", "--temporal_d_k", "16", "--temporal_d_v", "16", "--temporal_n_hidden", "64",
          "--temporal_MHA_layerNorm", "--temporal_FFN_layerNorm", "--temporal_relative_position_encoding", "--part_len", "3"]
    r = _run("Test", "evaluation_shanghaitech_ubnormal.py", ev + [
        "--dataset", "SHT", "--dataset_path", world["sht_feats"], "--testing_txt", world["sht_test"], "--test_mask_dir",
        world["sht_masks"], "--temporal_model_path", os.path.join(save, enc_ckpt), "--classifier_model_path",
        os.path.join(save, cls_ckpt)])
    assert r.returncode == 0, r.stderr[-2000:]
    auc = float(r.stdout.strip().split("auc = ")[-1])
    # the file name carries str(train AUC) (Train/temporal_transformer_shanghaitech.py:242-247); the TEST AUC of the epoch
    # that wrote the last checkpoint is in the log line printed right after the save
    train_auc = enc_ckpt.rsplit("_", 1)[-1]
    line = [l for l in train_log.splitlines() if "now test_AUC is" in l and ("now train_AUC is " + train_auc) in
            train_log[train_log.index(l):train_log.index(l) + len(l) + 200]]
    assert line, train_log[-1500:]
    assert abs(auc - float(line[-1].split("now test_AUC is ")[1].split()[0])) < 1e-3


@pytest.mark.parametrize("tag", ["ltn_sht", "stn_sht"])
def test_train_cli_evaluation_block_matches_reference_train(world, tmp_path, tag):
    """Train/temporal_transformer_shanghaitech.py / Train/spatio_transformer_shanghaitech.py through the HIP path, one epoch
    with both learning rates 0 from the world's checkpoints - the run the reference's own train() made for the fixture
    (tests/golden/make_golden_pipeline.py run_train_loops): same AUCs over the test and the training videos, same
    best-AUC log line, same save decision, same checkpoint file names up to the AUC's last digits."""
    save = str(tmp_path / "ck") + os.sep
    common = ["--dataset_path", world["sht_feats"], "--training_txt", world["sht_train"], "--testing_txt", world["sht_test"],
              "--test_mask_dir", world["sht_masks"], "--model_save_dir", save, "--batch_size", "2", "--part_num", "3",
              "--n_patch", "16", "--n_head", "2", "--d_model", "32", "--d_k", "16", "--d_v", "16", "--FFN_layerNorm",
              "--load_model", "--lr_encoder", "0", "--epochs", "1", "--inter_epoch", "1", "--seed", "3", "--save_threshold", "0.05",
              "--log_dir", str(tmp_path / "log")]
    if tag == "ltn_sht":
        script = "temporal_transformer_shanghaitech.py"
        extra = ["--part_len", "3", "--n_hidden", "64", "--MHA_layerNorm", "--relative_position_encoding", "--load_temporal_model_path",
                 world["ltn_sht_enc.ckpt"], "--load_classifier_model_path", world["ltn_sht_cls.ckpt"], "--lr_classifier", "0",
                 "--saved_prefix", "pre_"]
    else:
        script = "spatio_transformer_shanghaitech.py"
        extra = ["--part_len", "2", "--n_hidden", "47", "--train_dataset", world["sht_feats"], "--load_spatio_model_path",
                 world["stn_sht_enc.ckpt"], "--load_classifier_model_path", world["stn_sht_reg.ckpt"], "--lr_regressor", "0",
                 "--saved_prefix", ""]
    r = _run("Train", script, common + extra)
    assert r.returncode == 0, r.stderr[-2500:]
    ref_auc = G[f"tl_eval/{tag}/auc"]
    line = [l for l in r.stderr.splitlines() if "best_test_AUC" in l][-1]
    nxt = r.stderr[r.stderr.index(line):].splitlines()[1]
    got_test = float(line.split("now test_AUC is ")[1].split()[0])
    got_train = float(nxt.split("now train_AUC is ")[1].split()[0])
    assert abs(got_test - ref_auc[0]) < 1e-4 and abs(got_train - ref_auc[1]) < 1e-4
    want = G[f"tl_eval/{tag}/saved"].tolist()
    got = sorted(os.listdir(save))
    assert [g.rsplit("_", 1)[0] for g in got] == [w.rsplit("_", 1)[0] for w in want]
    assert all(abs(float(g.rsplit("_", 1)[1]) - float(w.rsplit("_", 1)[1])) < 1e-4 for g, w in zip(got, want))
    assert "saving model......" in r.stderr and "save complete." in r.stderr


@pytest.mark.parametrize("dtype", ["bf16", "f32x3"])
def test_coteaching_loop_chain(dtype, tmp_path):
    """BASELINE config 3 in miniature (STN -> pseudo labels -> LTN -> pseudo labels -> STN with MIL + BCE), five CLI
    processes chained by .npy files, in the bf16 GEMM mode the config is quoted in and in f32x3."""
    out = str(tmp_path / "coteach")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "coteach_loop_synthetic.sh"), out, dtype], capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT, HIP_VISIBLE_DEVICES="0"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    for f in ("STN_pseudo_labels.npy", "LTN_pseudo_labels.npy"):
        d = np.load(os.path.join(out, f), allow_pickle=True).tolist()
        assert len(d) == 16 and all(v.ndim == 2 and v.shape[1] == 1 and np.isfinite(v).all() for v in d.values())
    assert "spatio_loss" in r.stdout + r.stderr


def _losses_of(stderr, key, fields):
    rows = []
    for line in stderr.splitlines():
        if key in line and "]: " in line:
            body = line.split("]: ", 1)[1].replace(",", " ").split()
            vals = dict(zip(body[0::2], body[1::2]))
            rows.append([float(vals[f]) for f in fields])
    return np.array(rows)


def _run_chain(world, tmp_path, compute_dtype, feed=None):
    """The five co-teaching stages of BASELINE config 3 (Train/spatio_transformer_shanghaitech.py ->
    pseudo_labels_generator_spatio.py -> Train/temporal_transformer_shanghaitech.py -> pseudo_labels_generator_temporal.py ->
    Train/spatio_transformer_MIL_CE.py) as five command lines chained through this run's OWN files, in GEMM mode
    ``compute_dtype`` (--compute_dtype of every script), plus Test/evaluation_shanghaitech_ubnormal.py on the trained LTN.
    Returns the parsed loss rows, the checkpoint prefixes, the three pseudo-label dicts and the LTN's test AUC.
    ``feed``: directory of ANOTHER run whose intermediate files (trained checkpoints, label files) every stage reads instead of
    this run's own - each stage then starts from identical inputs in both runs ("teacher-forced" comparison of two modes)."""
    tmp_path.mkdir(parents=True, exist_ok=True)
    src = tmp_path if feed is None else feed
    save = str(tmp_path / "ck") + os.sep
    os.makedirs(save)
    strip = lambda src, dst: (torch.save({(k[7:] if k.startswith("module.") else k): v for k, v in torch.load(src).items()}, dst), dst)[1]
    thr_s, thr_t = (float(x) for x in G["chain/thr"])
    P = dict(world, save=save, ltn_enc_in=strip(world["ltn_sht_enc.ckpt"], str(tmp_path / "enc_l.ckpt")),
             ltn_cls_in=strip(world["ltn_sht_cls.ckpt"], str(tmp_path / "cls_l.ckpt")),
             stn_enc_out=str(src / "stn_encoder.ckpt"), stn_reg_out=str(src / "stn_head.ckpt"),
             pl_s=str(src / "pl_s.npy"), pl_t=str(src / "pl_t.npy"), pl_mce=str(tmp_path / "pl_mce"), thr_s=repr(thr_s))
    own = dict(pl_s=str(tmp_path / "pl_s.npy"), pl_t=str(tmp_path / "pl_t.npy"))      # what THIS run writes
    dt = ["--compute_dtype", compute_dtype]
    res = {"dir": tmp_path, "P": P}
    # 1. STN
    r = _run("Train", "spatio_transformer_shanghaitech.py", chain_argv("stn", P) + dt + ["--save_final", str(tmp_path / "stn_"), "--log_dir", str(tmp_path / "l1")])
    assert r.returncode == 0, r.stderr[-2500:]
    res["stn"] = _losses_of(r.stderr, "err", ["loss", "err", "l1"])
    # 2. its pseudo labels
    gen = ["--dataset_path", world["sht_feats"], "--training_txt", world["sht_train"], "--n_patch", "16", "--n_head", "2", "--d_model", "32",
           "--d_k", "16", "--d_v", "16", "--FFN_layerNorm"] + dt
    r = _run("Train", "pseudo_labels_generator_spatio.py", gen + ["--n_hidden", "47", "--threshold", repr(thr_s), "--spatio_model_path", P["stn_enc_out"],
                                                                  "--regression_model_path", P["stn_reg_out"], "--pseudo_labels_path", own["pl_s"]])
    assert r.returncode == 0, r.stderr[-2500:]
    res["pl_s"] = np.load(own["pl_s"], allow_pickle=True).tolist()
    # 3. LTN on those labels
    r = _run("Train", "temporal_transformer_shanghaitech.py", chain_argv("ltn", P) + dt + ["--save_final", str(tmp_path / "ltn_"), "--log_dir", str(tmp_path / "l3")])
    assert r.returncode == 0, r.stderr[-2500:]
    res["ltn"] = _losses_of(r.stderr, "MIL_l1", ["CE_loss", "MIL_loss", "MIL_l1"])
    # 4. its pseudo labels
    r = _run("Train", "pseudo_labels_generator_temporal.py", gen + ["--n_hidden", "64", "--part_len", "3", "--MHA_layerNorm", "--relative_position_encoding",
                                                                    "--threshold", repr(thr_t), "--temporal_model_path", str(src / "ltn_encoder.ckpt"),
                                                                    "--classifier_model_path", str(src / "ltn_head.ckpt"), "--pseudo_labels_path", own["pl_t"]])
    assert r.returncode == 0, r.stderr[-2500:]
    res["pl_t"] = np.load(own["pl_t"], allow_pickle=True).tolist()
    # 5. STN co-teaching on the LTN's labels + the end-of-round label file
    r = _run("Train", "spatio_transformer_MIL_CE.py", chain_argv("mce", P) + dt + ["--save_final", str(tmp_path / "mce_"), "--log_dir", str(tmp_path / "l5")])
    assert r.returncode == 0, r.stderr[-2500:]
    res["mce"] = _losses_of(r.stderr, "spatio_loss", ["MIL_loss", "err", "l1", "CE_loss"])
    assert "temporal pseudo label generation finished." in r.stderr
    res["pl_mce"] = np.load(P["pl_mce"] + ".npy", allow_pickle=True).tolist()
    # the trained LTN on the test videos (Test/evaluation_shanghaitech_ubnormal.py, the script README.md:50 runs)
    ev = ["--d_model", "32", "--temporal_n_head", "2", "--temporal_d_k", "16", "--temporal_d_v", "16", "--temporal_n_hidden", "64",
          "--temporal_MHA_layerNorm", "--temporal_FFN_layerNorm", "--temporal_relative_position_encoding", "--part_len", "3"] + dt
    r = _run("Test", "evaluation_shanghaitech_ubnormal.py", ev + [
        "--dataset", "SHT", "--dataset_path", world["sht_feats"], "--testing_txt", world["sht_test"], "--test_mask_dir",
        world["sht_masks"], "--temporal_model_path", str(src / "ltn_encoder.ckpt"), "--classifier_model_path",
        str(src / "ltn_head.ckpt")])
    assert r.returncode == 0, r.stderr[-2000:]
    res["ltn_test_auc"] = float(r.stdout.strip().split("auc = ")[-1])
    return res


@pytest.fixture(scope="module")
def chain_fp32(world, tmp_path_factory):
    return _run_chain(world, tmp_path_factory.mktemp("chain_fp32") / "run", "fp32")


def test_coteaching_chain_matches_the_reference_run_stage_by_stage(world, chain_fp32):
    """BASELINE config 3 in miniature against the reference's own run of the same five stages on the same world
    (tests/golden/make_golden_pipeline.py run_coteach_chain): Train/spatio_transformer_shanghaitech.py ->
    pseudo_labels_generator_spatio.py -> Train/temporal_transformer_shanghaitech.py on those labels ->
    pseudo_labels_generator_temporal.py -> Train/spatio_transformer_MIL_CE.py (MIL + BCE, then its end-of-round labels).
    The build's stages are chained through the build's OWN files (weights, label files), so differences accumulate: six
    optimisation steps per training stage at lr 1e-3 / 2e-3, fp32.  Bars: every step's loss terms 1e-3 (the log prints 4
    digits), trained-weight norms 1e-3 relative, pseudo-label files 2e-3 with the same zero pattern (thresholds sit in gaps)."""
    R, tmp_path = chain_fp32, chain_fp32["dir"]

    def check_weights(tag, prefix):
        for n, f in (("enc", "encoder.ckpt"), ("head", "head.ckpt")):
            norms, samp = weight_fingerprint(torch.load(prefix + f, map_location="cpu"))
            ref_n, ref_s = G[f"chain/{tag}/{n}_norm"], G[f"chain/{tag}/{n}_samp"]
            assert norms.shape == ref_n.shape
            assert np.all(np.abs(norms - ref_n) <= 1e-3 * np.abs(ref_n) + 1e-6), (tag, n, np.abs(norms - ref_n).max())
            # Adagrad's early updates are lr * sign-like: an entry whose gradient is at rounding level may move the other way
            assert np.mean(np.abs(samp - ref_s) < 2e-4) > 0.98 and np.abs(samp - ref_s).max() < 12 * float(CHAIN["lr_encoder"]) * 2, \
                (tag, n, np.mean(np.abs(samp - ref_s) < 2e-4), np.abs(samp - ref_s).max())

    def check_labels(out, prefix):
        keys = [k[len(prefix):] for k in G.files if k.startswith(prefix)]
        assert list(out.keys()) == keys
        for k in keys:
            ref, got = G[prefix + k], np.asarray(out[k], np.float32)
            assert got.shape == ref.shape and np.array_equal(got > 0, ref > 0), k
            assert np.abs(got - ref).max() < 2e-3, (k, np.abs(got - ref).max())

    got = R["stn"]
    assert got.shape == G["chain/stn/losses"].shape and np.abs(got - G["chain/stn/losses"]).max() < 1e-3, (got, G["chain/stn/losses"])
    check_weights("stn", str(tmp_path / "stn_"))
    check_labels(R["pl_s"], "chain/pl_s/")
    got, ref = R["ltn"], G["chain/ltn/losses"]                          # reference rows: [CE, MIL loss, err, l1]
    assert got.shape[0] == ref.shape[0] and np.abs(got - ref[:, [0, 1, 3]]).max() < 1e-3, (got, ref)
    check_weights("ltn", str(tmp_path / "ltn_"))
    check_labels(R["pl_t"], "chain/pl_t/")
    got, ref = R["mce"], G["chain/mce/losses"]                          # [MIL loss, err, l1, BCE]
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-3, (got, ref)
    check_weights("mce", str(tmp_path / "mce_"))
    check_labels(R["pl_mce"], "chain/pl_mce/")


def test_coteaching_chain_in_bf16_tracks_the_fp32_chain(world, chain_fp32, tmp_path):
    """BASELINE config 3 is quoted in bf16: the SAME five command lines (+ the Test/ script) with --compute_dtype bf16 (bf16
    MFMA products; f32 storage / accumulation / softmax / LayerNorm / loss / Adagrad) against the fp32 chain of the test above,
    which is pinned stage by stage to the reference's own run.  Every bf16 stage reads the fp32 chain's intermediate files
    (trained checkpoints, label files), so both modes start each stage from identical inputs and six optimisation steps
    measure the arithmetic - not the dynamics: left to feed on its own files the bf16 chain departs from the fp32 one after
    the first Adagrad steps exactly like two fp32 implementations do (Adagrad's first updates are lr*sign(g) and the MIL loss
    back-propagates through an arg-max over nearly tied bag scores; tools/coteach_modes_compare.py prints both trajectories).
    Bars: every logged loss term of every step within 5e-2 (measured 1.6e-2), pseudo-label files with the same zero pattern on
    >= 98 % of their entries (measured: all) and values within 2e-2 (measured 2.5e-3), test AUC of the trained LTN within 1e-2;
    and the bf16 numbers must differ from the fp32 ones (the mode really ran)."""
    B, A = _run_chain(world, tmp_path / "run", "bf16", feed=chain_fp32["dir"]), chain_fp32
    differs = 0.0
    for stage in ("stn", "ltn", "mce"):
        assert B[stage].shape == A[stage].shape and B[stage].size > 0 and np.isfinite(B[stage]).all()
        d = np.abs(B[stage] - A[stage])
        assert d.max() < 5e-2, (stage, d.max(1))
        differs = max(differs, float(d.max()))
    for f in ("pl_s", "pl_t", "pl_mce"):
        assert list(B[f].keys()) == list(A[f].keys())
        same = total = 0
        for k in A[f]:
            a, b = np.asarray(A[f][k], np.float32), np.asarray(B[f][k], np.float32)
            assert a.shape == b.shape and np.isfinite(b).all()
            same += int(((a > 0) == (b > 0)).sum()); total += a.size
            both = (a > 0) & (b > 0)
            if both.any():
                assert np.abs(a - b)[both].max() < 2e-2, (f, k, np.abs(a - b)[both].max())
                differs = max(differs, float(np.abs(a - b)[both].max()))
        assert same >= 0.98 * total, (f, same, total)
    assert abs(B["ltn_test_auc"] - A["ltn_test_auc"]) < 1e-2, (B["ltn_test_auc"], A["ltn_test_auc"])
    assert differs > 1e-5, "bf16 chain equals the fp32 chain to the last digit: the mode did not run"


def test_data_parallel_command_line_starts_ranks_and_shards_evaluation_and_labels(world, tmp_path):
    """``python Train/<script>.py --data_parallel --gpu 0,0`` on the one-GPU test box (LSTC_SHARE_DEVICE=1: both ranks on device
    0; RCCL refuses that, so LSTC_DIST_BACKEND=gloo carries the collectives - everything else is the multi-GPU job): the script
    starts its two ranks ITSELF (lstc_vad_amd.launch), every rank evaluates its share of the test and training videos
    (pipeline._ScoreBoard) and rank 0 logs / saves.  With both learning rates 0 the weights never move, so the AUC lines, the save
    decision and the checkpoint names must equal the single-process run CHARACTER FOR CHARACTER; the sharded pseudo-label
    generator must write a label file equal to the single-rank one bit for bit."""
    def cmd(save, log, extra):
        return ["--dataset_path", world["sht_feats"], "--training_txt", world["sht_train"], "--testing_txt", world["sht_test"],
                "--test_mask_dir", world["sht_masks"], "--model_save_dir", save, "--batch_size", "2", "--part_num", "3",
                "--n_patch", "16", "--n_head", "2", "--d_model", "32", "--d_k", "16", "--d_v", "16", "--FFN_layerNorm",
                "--load_model", "--lr_encoder", "0", "--epochs", "2", "--inter_epoch", "1", "--seed", "3", "--save_threshold", "0.05",
                "--log_dir", log, "--part_len", "3", "--n_hidden", "64", "--MHA_layerNorm", "--relative_position_encoding",
                "--load_temporal_model_path", world["ltn_sht_enc.ckpt"], "--load_classifier_model_path", world["ltn_sht_cls.ckpt"],
                "--lr_classifier", "0", "--saved_prefix", "pre_"] + extra
    env2 = dict(os.environ, PYTHONPATH=ROOT, LSTC_SHARE_DEVICE="1", LSTC_DIST_BACKEND="gloo", LSTC_RANK_TIMEOUT_S="500")
    env2.pop("HIP_VISIBLE_DEVICES", None)
    save1, save2 = str(tmp_path / "ck1") + os.sep, str(tmp_path / "ck2") + os.sep
    r1 = _run("Train", "temporal_transformer_shanghaitech.py", cmd(save1, str(tmp_path / "log1"), []))
    assert r1.returncode == 0, r1.stderr[-2500:]
    r2 = subprocess.run([sys.executable, "temporal_transformer_shanghaitech.py"] + cmd(save2, str(tmp_path / "log2"), ["--data_parallel", "--gpu", "0,0"]),
                        cwd=os.path.join(ROOT, "Train"), env=env2, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2500:]
    assert "--data_parallel: 2 rank(s)" in r2.stderr

    def auc_lines(err):
        return [l.split(": ", 1)[1] for l in err.splitlines() if "_AUC" in l and ": " in l]
    assert auc_lines(r1.stderr) and auc_lines(r1.stderr) == auc_lines(r2.stderr), (auc_lines(r1.stderr), auc_lines(r2.stderr))
    assert sorted(os.listdir(save1)) == sorted(os.listdir(save2)) and len(os.listdir(save1)) >= 2
    assert r2.stderr.count("saving model......") == r1.stderr.count("saving model......")       # rank 0 alone saves
    # the generator: two ranks score the training videos, rank 0 writes the file
    gen = ["--d_model", "32", "--n_head", "2", "--d_k", "16", "--d_v", "16", "--n_hidden", "64", "--MHA_layerNorm", "--FFN_layerNorm",
           "--relative_position_encoding", "--part_len", "3", "--dataset", "SHT", "--dataset_path", world["sht_feats"],
           "--training_txt", world["sht_train"], "--temporal_model_path", world["ltn_sht_enc.ckpt"], "--classifier_model_path",
           world["ltn_sht_cls.ckpt"], "--threshold", "0.45"]
    p1, p2 = str(tmp_path / "pl1.npy"), str(tmp_path / "pl2.npy")
    g1 = _run("Train", "pseudo_labels_generator_temporal.py", gen + ["--pseudo_labels_path", p1])
    assert g1.returncode == 0, g1.stderr[-2000:]
    g2 = subprocess.run([sys.executable, "pseudo_labels_generator_temporal.py"] + gen + ["--pseudo_labels_path", p2, "--data_parallel", "--gpu", "0,0"],
                        cwd=os.path.join(ROOT, "Train"), env=env2, capture_output=True, text=True, timeout=900)
    assert g2.returncode == 0, g2.stderr[-2500:]
    a, b = np.load(p1, allow_pickle=True).tolist(), np.load(p2, allow_pickle=True).tolist()
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    _check_pseudo(b, "pl/t_sht/", 0.45)


def _wide_sht_world(root):
    """A larger SHT-dialect set than the golden-pinned world (which has 4 abnormal training videos): 18 normal + 17 abnormal training
    videos and 9 test videos, P = 16, d = 32 - enough for a global batch of 16 pairs = 8 ranks x 2 pairs and for an 8-way sharded
    evaluation (video i on rank i % 8).  Not a fixture of the reference: the 8-rank run is compared with the single-process run."""
    from lstc_vad_amd.archive import write_archive
    os.makedirs(root, exist_ok=True)
    train = [(f"{1 + i % 9:02d}_{100 + i:04d}" if i % 2 else f"{1 + i % 9:02d}_{100 + i:03d}", i % 2, 9 + (7 * i) % 31) for i in range(35)]
    test = [(f"{1 + i % 5:02d}_{500 + i:04d}" if i % 2 else f"{1 + i % 5:02d}_{500 + i:03d}", i % 2, 5 + (5 * i) % 17) for i in range(9)]
    arrays = {n + ".npy": pw._feats(c, 16, 32, 700 + i) for i, (n, _, c) in enumerate(train + test)}
    W = {"feats": write_archive(os.path.join(root, "wide_feats.npz"), arrays)}
    W["train"] = os.path.join(root, "wide_train.txt")
    open(W["train"], "w").write("".join(f"{n},{l}\n" for n, l, _ in train))
    W["test"] = os.path.join(root, "wide_test.txt")
    open(W["test"], "w").write("".join(f"{n},{l},{-1 if l else c * pw.SEG + 3}\n" for n, l, c in test))
    W["masks"] = os.path.join(root, "wide_masks") + os.sep
    os.makedirs(W["masks"], exist_ok=True)
    for i, (n, l, c) in enumerate(test + train):
        if l:
            np.save(os.path.join(W["masks"], n + ".npy"), pw._mask(c * pw.SEG + 5, 800 + i))
    return W


@pytest.mark.timeout(2400)
def test_data_parallel_command_line_with_eight_ranks(world, tmp_path):
    """``python Train/temporal_transformer_shanghaitech.py --data_parallel --gpu 0,0,0,0,0,0,0,0`` (LSTC_SHARE_DEVICE=1, gloo): the
    script starts EIGHT ranks itself; a global batch of 16 pairs = 2 pairs per rank; evaluation, train-AUC and (second command) the
    pseudo-label generator sharded ``i % 8`` over the ranks.  (a) with both learning rates 0: AUC lines, save decisions and
    checkpoint names equal the single-process run character for character, label file bit for bit; (b) with the reference's
    learning rates: every logged step's loss terms agree with the single-process run's to 1e-4 (the 8 ranks' gradients summed to
    the batch gradient, step after step) and the AUCs after training to 1e-4."""
    W = _wide_sht_world(str(tmp_path / "wide"))

    def cmd(save, log, extra):
        return ["--dataset_path", W["feats"], "--training_txt", W["train"], "--testing_txt", W["test"],
                "--test_mask_dir", W["masks"], "--model_save_dir", save, "--batch_size", "16", "--part_num", "3",
                "--n_patch", "16", "--n_head", "2", "--d_model", "32", "--d_k", "16", "--d_v", "16", "--FFN_layerNorm",
                "--load_model", "--epochs", "3", "--inter_epoch", "1", "--seed", "3", "--save_threshold", "0.05",
                "--log_dir", log, "--part_len", "3", "--n_hidden", "64", "--MHA_layerNorm", "--relative_position_encoding",
                "--load_temporal_model_path", world["ltn_sht_enc.ckpt"], "--load_classifier_model_path", world["ltn_sht_cls.ckpt"],
                "--MHA_attn_dropout", "0", "--MHA_fc_dropout", "0", "--FFN_dropout", "0", "--classifier_dropout", "0",
                "--saved_prefix", "pre_"] + extra
    env8 = dict(os.environ, PYTHONPATH=ROOT, LSTC_SHARE_DEVICE="1", LSTC_DIST_BACKEND="gloo", LSTC_RANK_TIMEOUT_S="1500")
    env8.pop("HIP_VISIBLE_DEVICES", None)
    eight = ["--data_parallel", "--gpu", "0,0,0,0,0,0,0,0"]

    def auc_lines(err):
        return [l.split(": ", 1)[1] for l in err.splitlines() if "_AUC" in l and ": " in l]

    def numbers(line):
        import re
        return [float(x) for x in re.findall(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", line)]
    for tag, lrs in (("frozen", ["--lr_encoder", "0", "--lr_classifier", "0"]), ("training", [])):
        save1, save8 = str(tmp_path / f"{tag}_ck1") + os.sep, str(tmp_path / f"{tag}_ck8") + os.sep
        r1 = _run("Train", "temporal_transformer_shanghaitech.py", cmd(save1, str(tmp_path / f"{tag}_log1"), lrs))
        assert r1.returncode == 0, r1.stderr[-2500:]
        r8 = subprocess.run([sys.executable, "temporal_transformer_shanghaitech.py"] + cmd(save8, str(tmp_path / f"{tag}_log8"), lrs + eight),
                            cwd=os.path.join(ROOT, "Train"), env=env8, capture_output=True, text=True, timeout=2000)
        assert r8.returncode == 0, r8.stderr[-2500:]
        assert "--data_parallel: 8 rank(s)" in r8.stderr
        a1, a8 = auc_lines(r1.stderr), auc_lines(r8.stderr)
        assert a1 and len(a1) == len(a8)
        if tag == "frozen":
            assert a1 == a8, (a1, a8)
            assert sorted(os.listdir(save1)) == sorted(os.listdir(save8)) and len(os.listdir(save1)) >= 2
            assert r8.stderr.count("saving model......") == r1.stderr.count("saving model......")
        else:
            for x, y in zip(a1, a8):
                nx, ny = numbers(x), numbers(y)
                assert len(nx) == len(ny) and nx and max(abs(p - q) for p, q in zip(nx, ny)) < 2e-4, (x, y)     # 4-decimal log format
            # the per-step log lines (loss terms of every optimisation step, rank 0): same count, same numbers to 1e-4
            s1 = [l for l in r1.stderr.splitlines() if "loss" in l.lower() and "_AUC" not in l]
            s8 = [l for l in r8.stderr.splitlines() if "loss" in l.lower() and "_AUC" not in l and "rank" not in l.lower()]
            assert s1 and len(s1) == len(s8), (len(s1), len(s8))
            for x, y in zip(s1, s8):
                nx, ny = numbers(x), numbers(y)
                assert len(nx) == len(ny) and max(abs(p - q) for p, q in zip(nx, ny)) < 2e-4, (x, y)
    gen = ["--d_model", "32", "--n_head", "2", "--d_k", "16", "--d_v", "16", "--n_hidden", "64", "--MHA_layerNorm", "--FFN_layerNorm",
           "--relative_position_encoding", "--part_len", "3", "--dataset", "SHT", "--dataset_path", W["feats"],
           "--training_txt", W["train"], "--temporal_model_path", world["ltn_sht_enc.ckpt"], "--classifier_model_path",
           world["ltn_sht_cls.ckpt"], "--threshold", "0.45"]
    p1, p8 = str(tmp_path / "pl1.npy"), str(tmp_path / "pl8.npy")
    g1 = _run("Train", "pseudo_labels_generator_temporal.py", gen + ["--pseudo_labels_path", p1])
    assert g1.returncode == 0, g1.stderr[-2000:]
    g8 = subprocess.run([sys.executable, "pseudo_labels_generator_temporal.py"] + gen + ["--pseudo_labels_path", p8] + eight,
                        cwd=os.path.join(ROOT, "Train"), env=env8, capture_output=True, text=True, timeout=2000)
    assert g8.returncode == 0, g8.stderr[-2500:]
    a, b = np.load(p1, allow_pickle=True).tolist(), np.load(p8, allow_pickle=True).tolist()
    assert len(a) == 35 and list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)


def test_sharded_evaluation_is_bitwise_the_single_rank_evaluation(world):
    """pipeline.evaluate_auc / evaluate_train_auc / generate_pseudo_labels with two emulated ranks on the HIP model (fp32): every
    frame score, label, AUC and label row equals the single-rank pass BIT FOR BIT - a sequence's score does not depend on what
    shares its batch, and the exchange is a zero-padded sum."""
    enc, head = _load(world, "ltn_sht_enc.ckpt", pw.LTN_SHT, "ltn_sht_cls.ckpt", Classifier)

    def two(run):
        kept = {}
        run(1, 2, lambda flat: kept.__setitem__(1, flat.clone()))
        return run(0, 2, lambda flat: flat.add_(kept[1]))
    ev = lambda r=None, w=None, ex=None: pipeline.evaluate_auc(enc, head, "LTN", "SHT", world["sht_feats"], world["sht_test"], world["sht_masks"],
                                                               3, 16, return_frames=True, rank=r, world=w, exchange=ex, pool_sequences=5)
    a1, a2 = ev(), two(ev)
    assert a1[0] == a2[0] and np.array_equal(a1[1], a2[1]) and np.array_equal(a1[2], a2[2])
    tr = lambda r=None, w=None, ex=None: pipeline.evaluate_train_auc(enc, head, "LTN", "SHT", world["sht_feats"], world["sht_train"],
                                                                     world["sht_masks"], 3, 16, return_frames=True, rank=r, world=w, exchange=ex)
    b1, b2 = tr(), two(tr)
    assert b1[0] == b2[0] and np.array_equal(b1[1], b2[1]) and np.array_equal(b1[2], b2[2])
    pl = lambda r=None, w=None, ex=None: pipeline.generate_pseudo_labels(enc, head, "LTN", "SHT", world["sht_feats"], world["sht_train"], 0.45,
                                                                         part_len=3, rank=r, world=w, exchange=ex)
    c1, c2 = pl(), two(pl)
    assert list(c1) == list(c2) and all(np.array_equal(c1[k], c2[k]) for k in c1)
