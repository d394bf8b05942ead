"""CPU tests of the stages around the training step: dataset classes / test loaders against fixtures made
by the reference's own classes (tests/golden/make_golden_pipeline.py), archive backends, and the host logic of
pseudo-label generation / evaluation with the ORACLE standing in for the device model (the product path has no CPU
model; here only the file formats, part arithmetic, thresholds and frame expansion are under test)."""
import os
import random
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import pipeline_world as pw                                   # noqa: E402
from pipeline_cases import DATASET_CASES, build_dataset       # noqa: E402
from lstc_vad_amd import load_dataset as ds_mod               # noqa: E402
from lstc_vad_amd import pipeline                             # noqa: E402
from lstc_vad_amd.archive import FeatureArchive               # noqa: E402
from lstc_vad_amd.models import Classifier, Encoder, Regressor  # noqa: E402
from oracle import lstc_oracle as orc                         # noqa: E402

G = np.load(os.path.join(HERE, "golden", "pipeline.npz"))


H5DIR = os.path.join(HERE, "golden", "hdf5")


@pytest.fixture(scope="module")
def world_npz(tmp_path_factory):
    return pw.build(str(tmp_path_factory.mktemp("world")), Encoder, Regressor, Classifier)


@pytest.fixture(scope="module", params=["npz", "hdf5"])
def world(request, world_npz):
    """The synthetic world with its feature archives as numpy files, and again with the SHT / UCF feature archives and the UCF
    ground truth as REAL HDF5 files (written by libhdf5: tests/golden/make_hdf5_fixtures.py) read by lstc_vad_amd.hdf5 - the
    container the reference's datasets actually come in."""
    if request.param == "npz":
        return world_npz
    W = dict(world_npz)
    for tag in ("sht_feats", "ucf_feats", "ucf_gt"):
        W[tag] = os.path.join(H5DIR, f"world_{tag}.h5")
    return W


def _fingerprint(t):
    a = t.numpy()
    return a.reshape(a.shape[0], -1)[:, 0].copy(), np.array([a.astype(np.float64).sum()])


@pytest.mark.parametrize("name", list(DATASET_CASES))
def test_dataset_classes_sample_like_the_reference(world, name):
    """Same seeds -> the same clips, labels, crops and shapes as utils/load_dataset.py, item by item over two epochs."""
    spec = DATASET_CASES[name]
    np.random.seed(spec["seed"]); random.seed(spec["seed"])
    ds = build_dataset(ds_mod, spec, world)
    assert len(ds) == int(G[f"ds/{name}/len"][0])
    firsts, sums, labs, crops = [], [], [], []
    for _ in range(2):
        for i in range(len(ds)):
            item = ds[i]
            for j in (0, 2):
                f, s = _fingerprint(item[j]); firsts.append(f); sums.append(s)
                labs.append(item[j + 1].numpy().reshape(-1))
                assert item[j].dtype == torch.float32 and item[j + 1].dtype == torch.float32
            if len(item) == 5:
                crops.append(item[4])
        ds.shuffle_keys()
    assert np.array_equal(np.concatenate(firsts), G[f"ds/{name}/first"])
    assert np.array_equal(np.concatenate(sums), G[f"ds/{name}/sum"])
    assert np.array_equal(np.concatenate(labs), G[f"ds/{name}/labs"])
    assert np.array_equal(np.array(item[0].shape), G[f"ds/{name}/shape"])
    assert np.array_equal(np.array(item[1].shape), G[f"ds/{name}/lab_shape"])
    assert np.array_equal(np.array(crops, np.int64), G[f"ds/{name}/crops"])


def test_test_loaders_match_reference(world):
    for tag, fn, args in (("sht", ds_mod.shanghaitech_test, (world["sht_test"], world["sht_masks"], world["sht_feats"])),
                          ("ubn", ds_mod.UBnormal_test, (world["ubn_test"], world["ubn_masks"], world["ubn_feats"]))):
        feats, labels, annos, names = fn(*args, return_names=True)
        assert np.array_equal([f.shape[0] for f in feats], G[f"tl/{tag}/n_clips"])
        assert np.array_equal([l == "Abnormal" for l in labels], G[f"tl/{tag}/abnormal"])
        assert np.array_equal([len(a) for a in annos], G[f"tl/{tag}/anno_len"])
        assert np.array_equal([float(np.sum(a)) for a in annos], G[f"tl/{tag}/anno_sum"])
        assert len(fn(*args)) == 3
    for i, line in enumerate(open(world["ucf_test"]).readlines()):
        feats, anno, n_frames, key = ds_mod.UCF_test(line, world["ucf_feats"], world["ucf_gt"], 16, return_name=True)
        assert np.array_equal([feats.shape[0], len(anno), float(np.sum(anno)), n_frames], G[f"tl/ucf/{i}"])


def test_archive_backends_agree(world):
    with FeatureArchive(world["sht_feats"]) as a, FeatureArchive(world["sht_feats_dir"]) as b:
        assert sorted(a.keys()) == sorted(b.keys())
        for k in a.keys():
            assert k in b and np.array_equal(np.asarray(a[k]), np.asarray(b[k]))
    with pytest.raises(FileNotFoundError):
        FeatureArchive(os.path.join(world["root"], "missing.npz"))
    fake = os.path.join(world["root"], "fake.h5")
    open(fake, "wb").write(b"\x89HDF\r\n\x1a\n" + b"\xff" * 64)          # signature, then nothing a superblock could be
    with pytest.raises(OSError):
        FeatureArchive(fake)


def test_missing_pseudo_label_file_exits_like_reference(world):
    with pytest.raises(SystemExit):
        ds_mod.SH_Train_Origin_Dataset(3, 2, world["sht_feats"], world["sht_train"], 16, "uniform",
                                       pseudo_labels_path=os.path.join(world["root"], "nope.npy"))


# ---- host logic of the stages, oracle as the model -------------------------------------------------------------------

class _OracleEncoder:
    """Duck-typed stand-in for ``lstc_vad_amd.models.Encoder`` on CPU (tests only)."""

    def __init__(self, ckpt, kw):
        sd = torch.load(ckpt, map_location="cpu")
        self.P = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        kw = dict(kw); kw.pop("n_layers")
        self.cfg = orc.EncoderCfg(n_layers=3, **kw)
        self.layer_norm = types.SimpleNamespace(normalized_shape=(kw["d_model"],))

    def parameters(self):
        yield torch.zeros(1)

    def forward_cls(self, x):
        return orc.encoder_forward(self.P, x, self.cfg, training=False)[:, 0, :]


class _OracleHead:
    def __init__(self, ckpt, kind):
        sd = torch.load(ckpt, map_location="cpu")
        self.P = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.kind = kind

    def __call__(self, feats):
        return orc.head_forward(self.P, feats, self.kind, training=False)


def _check_pseudo(out, prefix, thr):
    keys = [k[len(prefix):] for k in G.files if k.startswith(prefix)]
    assert list(out.keys()) == keys                       # insertion order = list-file order, as np.save pickles it
    for k in keys:
        ref, got = G[prefix + k], out[k]
        assert got.shape == ref.shape and got.dtype == np.float32
        near = np.abs(np.where(ref > 0, ref, got) - thr) < 1e-5       # a score sitting on the threshold may flip
        assert np.all(np.abs(got - ref)[~near] < 2e-6), k
        assert (ref > 0).any() or (ref == 0).all()


def test_pseudo_label_generation_host_logic(world, tmp_path):
    enc = _OracleEncoder(world["ltn_sht_enc.ckpt"], pw.LTN_SHT); head = _OracleHead(world["ltn_sht_cls.ckpt"], "classifier")
    p = str(tmp_path / "pl.npy")
    out = pipeline.generate_pseudo_labels(enc, head, "LTN", "SHT", world["sht_feats"], world["sht_train"], 0.45, part_len=3,
                                          out_path=p)
    _check_pseudo(out, "pl/t_sht/", 0.45)
    back = np.load(p, allow_pickle=True).tolist()         # the reference's reader (utils/load_dataset.py:20)
    assert list(back) == list(out) and all(np.array_equal(back[k], out[k]) for k in out)
    out = pipeline.generate_pseudo_labels(enc, head, "LTN", "UBnormal", world["ubn_feats"], world["ubn_train"], 0.45, part_len=3)
    _check_pseudo(out, "pl/t_ubn/", 0.45)
    enc = _OracleEncoder(world["ltn_ucf_enc.ckpt"], pw.LTN_UCF); head = _OracleHead(world["ltn_ucf_cls.ckpt"], "classifier")
    out = pipeline.generate_pseudo_labels(enc, head, "LTN", "UCF", world["ucf_feats"], world["ucf_train"], 0.25, part_len=2,
                                          n_patch=9)
    _check_pseudo(out, "pl/t_ucf/", 0.25)
    enc = _OracleEncoder(world["stn_sht_enc.ckpt"], pw.STN_SHT); head = _OracleHead(world["stn_sht_reg.ckpt"], "regressor")
    out = pipeline.generate_pseudo_labels(enc, head, "STN", "SHT", world["sht_feats"], world["sht_train"], 0.34)
    _check_pseudo(out, "pl/s_sht/", 0.34)


def test_evaluation_host_logic(world):
    enc = _OracleEncoder(world["ltn_sht_enc.ckpt"], pw.LTN_SHT); head = _OracleHead(world["ltn_sht_cls.ckpt"], "classifier")
    for tag, dataset, txt, masks, feats in (("sht", "SHT", "sht_test", "sht_masks", "sht_feats"),
                                             ("ubn", "UBnormal", "ubn_test", "ubn_masks", "ubn_feats")):
        auc, s, l = pipeline.evaluate_auc(enc, head, "LTN", dataset, world[feats], world[txt], world[masks], 3, 16,
                                          return_frames=True)
        assert s.shape == G[f"ev/{tag}/scores"].shape
        assert np.max(np.abs(s - G[f"ev/{tag}/scores"])) < 2e-6
        assert np.array_equal(l, G[f"ev/{tag}/labels"])
        assert abs(auc - float(G[f"ev/{tag}/auc"][0])) < 1e-9
    enc = _OracleEncoder(world["ltn_ucf_enc.ckpt"], pw.LTN_UCF); head = _OracleHead(world["ltn_ucf_cls.ckpt"], "classifier")
    auc, s, l = pipeline.evaluate_auc(enc, head, "LTN", "UCF", world["ucf_feats"], world["ucf_test"], world["ucf_gt"], 2, 9,
                                      return_frames=True)
    assert np.max(np.abs(s - G["ev/ucf/scores"])) < 2e-6 and np.array_equal(l, G["ev/ucf/labels"])
    assert abs(auc - float(G["ev/ucf/auc"][0])) < 1e-9


def test_utils_import_shim(world):
    """``from utils.load_dataset import ...`` / ``from utils.eval_utils import eval`` as the reference scripts spell it."""
    import utils.eval_utils as ue
    import utils.load_dataset as ul
    import utils.utils as uu
    assert ul.SH_Train_Origin_Dataset is ds_mod.SH_Train_Origin_Dataset and ul.UCF_test is ds_mod.UCF_test
    assert ue.eval([0.1, 0.9, 0.3], [0, 1, 0], None) == 1.0
    assert uu.get_video_names(world["sht_test"], abnormal=True, normal=False) == [n for n, l, _ in pw.SHT_TEST if l]


@pytest.mark.parametrize("name", ["sh_uniform", "sh_random_pseudo", "ubn_uniform"])
def test_rank_shards_are_disjoint_and_union_is_the_single_process_batch(world, name):
    """Data parallelism over videos: every rank seeds np.random identically and walks the same permutation
    and window draws; rank r keeps pairs [r*bs/G, (r+1)*bs/G).  The shards' union must equal the batch one process forms,
    which in turn is what iterating the dataset item by item gives (the reference's DataLoader order)."""
    spec = DATASET_CASES[name]
    bs = 2

    def plans(rank, world_size):
        np.random.seed(spec["seed"]); random.seed(spec["seed"])
        ds = build_dataset(ds_mod, spec, world)
        offs = np.concatenate([[0], np.cumsum([v.shape[0] for v in ds.norm_feats + ds.abnorm_feats])]).astype(np.int64)
        out = []
        for epoch in range(2):
            for b in range(len(ds) // bs):
                out.append(ds_mod.shard_plan(ds, offs, len(ds.norm_feats), b, bs, rank, world_size))
            ds.shuffle_keys()
        return out, ds, offs

    single, ds, offs = plans(0, 1)
    r0, _, _ = plans(0, 2)
    r1, _, _ = plans(1, 2)
    assert len(single) >= 2
    for (i1, l1), (ia, la), (ib, lb) in zip(single, r0, r1):
        assert np.array_equal(np.concatenate([ia, ib], 1), i1) and np.array_equal(np.concatenate([la, lb], 1), l1)
        assert ia.shape[1] == ib.shape[1] == bs // 2
    # and the single-process plan addresses exactly the clips that __getitem__ returns
    np.random.seed(spec["seed"]); random.seed(spec["seed"])
    ds2 = build_dataset(ds_mod, spec, world)
    bank = np.concatenate([v[:, :ds2.n_patch] if ds2.n_patch != 1 else v for v in ds2.norm_feats + ds2.abnorm_feats], 0)
    idx, labs = single[0]
    for j in range(bs):
        item = ds2[j]
        assert np.array_equal(bank[idx[0, j]], item[0].numpy()) and np.array_equal(bank[idx[1, j]], item[2].numpy())
        assert np.array_equal(labs[1, j].reshape(item[3].shape), item[3].numpy())
    with pytest.raises(ValueError):
        ds_mod.ResidentPairs(ds2, 3, "cpu", 0, 2)            # 3 pairs do not split over 2 ranks


def test_train_loop_evaluation_and_checkpoint_rule_like_the_reference_train(world):
    """The in-loop evaluation + selection of the SHT train scripts, pinned to the reference's own ``train(args)`` run on the
    world with learning rates 0 (tests/golden/make_golden_pipeline.py ``run_train_loops``): frame scores / labels of the
    test AND the training videos, both AUCs, the best-AUC log line, the save decision and the checkpoint file names
    (``<prefix>temporal_model_oneCrop_<type>_<str(auc_train)>``).  Host logic here (oracle as the model)."""
    from argparse import Namespace
    from lstc_vad_amd import cli
    for tag, script, enc_ck, head_ck, kw, kind, mode, L, prefix in (
            ("ltn_sht", "temporal_transformer_shanghaitech", "ltn_sht_enc.ckpt", "ltn_sht_cls.ckpt", pw.LTN_SHT, "classifier", "LTN", 3, "pre_"),
            ("stn_sht", "spatio_transformer_shanghaitech", "stn_sht_enc.ckpt", "stn_sht_reg.ckpt", pw.STN_SHT, "regressor", "STN", 2, "")):
        enc, head = _OracleEncoder(world[enc_ck], kw), _OracleHead(world[head_ck], kind)
        at, st, lt = pipeline.evaluate_auc(enc, head, mode, "SHT", world["sht_feats"], world["sht_test"], world["sht_masks"], L, 16,
                                           return_frames=True)
        ar, sr, lr = pipeline.evaluate_train_auc(enc, head, mode, "SHT", world["sht_feats"], world["sht_train"], world["sht_masks"],
                                                 L, 16, return_frames=True)
        for got, key in ((st, "test_scores"), (sr, "train_scores")):
            assert got.shape == G[f"tl_eval/{tag}/{key}"].shape and np.max(np.abs(got - G[f"tl_eval/{tag}/{key}"])) < 2e-6
        assert np.array_equal(lt, G[f"tl_eval/{tag}/test_labels"]) and np.array_equal(lr, G[f"tl_eval/{tag}/train_labels"])
        ref_auc = G[f"tl_eval/{tag}/auc"]
        assert abs(at - ref_auc[0]) < 1e-9 and abs(ar - ref_auc[1]) < 1e-9
        # selection rule + file names, fed with the reference's AUC values
        args = Namespace(save_threshold=0.05, saved_prefix=prefix, type="I3D_RGB", model_save_dir="")
        sel = cli.Selector(script, args)
        save, lines = sel.update(0, float(ref_auc[0]), float(ref_auc[1]))
        assert save == float(ref_auc[1])
        names = sorted(os.path.basename(p) for p in cli.checkpoint_names(script, args, save))
        assert names == G[f"tl_eval/{tag}/saved"].tolist()
        assert lines[0] == [l for l in G[f"tl_eval/{tag}/log"].tolist() if l.startswith("best_")][0]
        # second evaluation: no improvement -> no save; better train AUC under the threshold -> no save either
        assert sel.update(10, 0.9, float(ref_auc[1]))[0] is None and sel.best_test == 0.9 and sel.best_test_epoch == 10
        sel2 = cli.Selector(script, Namespace(save_threshold=0.99, saved_prefix=None, type="I3D_RGB", model_save_dir="/m/"))
        assert sel2.update(0, 0.5, 0.6)[0] is None and sel2.best_train == 0.6
    # the other scripts' rules (file:line in cli.SELECTION)
    a = Namespace(save_threshold=0.8, saved_prefix="p_", type="I3D_RGB", model_save_dir="/m/")
    s = cli.Selector("spatio_transformer_UCF", a)
    assert s.update(0, 0.7, 0.0)[0] is None and s.update(5, 0.85, 0.0)[0] == 0.85          # test AUC gates, threshold applies
    assert cli.checkpoint_names("spatio_transformer_UCF", a, 0.85) == ("/m/spatio_model_oneCrop_0.85", "/m/regression_model_oneCrop_0.85")
    assert cli.checkpoint_names("temporal_transformer_UCF", a, 0.85)[0] == "/m/temporal_model_oneCrop_I3D_RGB_0.85"
    assert cli.checkpoint_names("temporal_transformer_UBnormal", a, 0.5)[1] == "/m/classifier_model_oneCrop_I3D_RGB_0.5"
    assert cli.checkpoint_names("spatio_transformer_UBnormal", Namespace(type="T", model_save_dir="/m"), 0.5)[0] == "/m/spatio_model_oneCrop_T_0.5"
    s = cli.Selector("spatio_transformer_UBnormal", a)
    assert s.update(0, 0.9, 0.0)[0] is None                                               # auc_train = 0 upstream: never saves
    s = cli.Selector("spatio_transformer_MIL_CE", a)
    assert s.best_test == 0.8 and s.update(0, 0.5, 0.3)[0] == 0.3 and s.best_test == 0.8   # no threshold on the save, best_test starts at it
    assert cli.checkpoint_names("spatio_transformer_MIL_CE", a, 0.3)[0] == "/m/p_spatio_model_oneCrop_I3D_RGB_0.3"


@pytest.mark.parametrize("name", ["sh_uniform", "sh_random_pseudo", "ucf_uniform", "sh_tencrop_uniform"])
def test_loader_worker_rng_streams_like_the_reference_dataloader(world, name):
    """``load_dataset.WorkerStreams`` against the reference's dataset classes behind a REAL torch DataLoader with k worker
    processes and the scripts' ``worker_init`` (fixture ``dlw/*``): batch b comes from worker b % k, whose generators are
    seeded ``seed + worker`` at the start of EVERY epoch; ``shuffle_keys`` draws from the parent's generator."""
    spec = DATASET_CASES[name]
    k, bs = (int(x) for x in G[f"dlw/{name}/cfg"])
    seed = spec["seed"]
    np.random.seed(seed); random.seed(seed)
    ds = build_dataset(ds_mod, spec, world)
    streams = ds_mod.WorkerStreams(k, seed)
    firsts, labs, crops = [], [], []
    for epoch in range(2):
        streams.begin_epoch()
        for b in range(len(ds) // bs):
            with streams.batch(b):
                items = [ds[b * bs + j] for j in range(bs)]
            for j in (0, 2):
                a = torch.stack([it[j] for it in items]).numpy()
                firsts.append(a.reshape(a.shape[0] * a.shape[1], -1)[:, 0].copy())
                labs.append(torch.stack([it[j + 1] for it in items]).numpy().reshape(-1))
            if len(items[0]) == 5:
                crops.extend(int(it[4]) for it in items)
        ds.shuffle_keys()
    assert np.array_equal(np.concatenate(firsts), G[f"dlw/{name}/first"])
    assert np.array_equal(np.concatenate(labs), G[f"dlw/{name}/labs"])
    assert np.array_equal(np.array(crops, np.int64), G[f"dlw/{name}/crops"])
    if not ds.lazy:
        # the HBM-resident source plans the same clips: shard_plan under the same streams (ranks 0 and 1 of 2 when bs splits).
        # Round 6: the ten-crop classes too - the bank holds clip c's crop k in row 10 c + k and the per-item crop draw (Python's
        # ``random``, in the worker's stream) is part of the planned row
        np.random.seed(seed); random.seed(seed)
        ds2 = build_dataset(ds_mod, spec, world)
        st2 = ds_mod.WorkerStreams(k, seed)
        crops = 10 if ds2.ten_crop else 1
        vids = [v.reshape((-1,) + v.shape[2:]) if ds2.ten_crop else (v[:, :ds2.n_patch] if ds2.n_patch != 1 else v)
                for v in ds2.norm_feats + ds2.abnorm_feats]
        offs = np.concatenate([[0], np.cumsum([v.shape[0] for v in vids])]).astype(np.int64)
        bank = np.concatenate(vids, 0)
        got = []
        for epoch in range(2):
            st2.begin_epoch()
            for b in range(len(ds2) // bs):
                with st2.batch(b):
                    idx, _ = ds_mod.shard_plan(ds2, offs, len(ds2.norm_feats), b, bs, crops=crops)
                for kind in (0, 1):
                    a = bank[idx[kind].reshape(-1)]
                    got.append(a.reshape(a.shape[0], -1)[:, 0].copy())
            ds2.shuffle_keys()
        assert np.array_equal(np.concatenate(got), G[f"dlw/{name}/first"])
    # num_workers = 0 keeps the caller's generators (the order the ds/* fixtures pin)
    np.random.seed(seed); a0 = np.random.get_state()[1].copy()
    with ds_mod.WorkerStreams(0, seed).batch(3):
        pass
    assert np.array_equal(np.random.get_state()[1], a0)


# ---- the same stages sharded over ranks (round 5) ----------------------------------------------------------------------

def _emulate_two_ranks(run):
    """``run(rank, world, exchange)`` for both ranks of a two-rank pass in ONE process: rank 1's zero-padded score vector is kept
    and added into rank 0's, which is what the SUM all-reduce of pipeline._ScoreBoard leaves on every rank."""
    kept = {}
    run(1, 2, lambda flat: kept.__setitem__(1, flat.clone()))
    return run(0, 2, lambda flat: flat.add_(kept[1]))


def test_sharded_passes_equal_the_single_rank_passes(world, tmp_path):
    """pipeline.generate_pseudo_labels / evaluate_auc / evaluate_train_auc with the videos of the list dealt to two ranks
    (i % 2), every rank walking the whole list on the host and scoring only its own videos: keys, shapes, labels and zero
    patterns identical to the single-rank pass, scores equal (the oracle's CPU BLAS is the model here, so to 2e-6; the HIP model
    is batch-invariant and tests/test_pipeline_gpu.py requires bit equality).  Rank 0 alone writes the label file."""
    enc = _OracleEncoder(world["ltn_sht_enc.ckpt"], pw.LTN_SHT); head = _OracleHead(world["ltn_sht_cls.ckpt"], "classifier")
    one = pipeline.generate_pseudo_labels(enc, head, "LTN", "SHT", world["sht_feats"], world["sht_train"], 0.45, part_len=3)
    p0, p1 = str(tmp_path / "r0.npy"), str(tmp_path / "r1.npy")
    two = _emulate_two_ranks(lambda r, w, ex: pipeline.generate_pseudo_labels(
        enc, head, "LTN", "SHT", world["sht_feats"], world["sht_train"], 0.45, part_len=3, out_path=(p0 if r == 0 else p1),
        rank=r, world=w, exchange=ex, pool_sequences=7))
    assert os.path.exists(p0) and not os.path.exists(p1)
    assert list(one) == list(two)
    for k in one:
        assert one[k].shape == two[k].shape and two[k].dtype == np.float32
        near = np.abs(np.where(one[k] > 0, one[k], two[k]) - 0.45) < 1e-5
        assert np.all(np.abs(one[k] - two[k])[~near] < 2e-6), k
    _check_pseudo(two, "pl/t_sht/", 0.45)
    for tag, dataset, txt, masks, feats in (("sht", "SHT", "sht_test", "sht_masks", "sht_feats"),
                                             ("ubn", "UBnormal", "ubn_test", "ubn_masks", "ubn_feats")):
        auc, s, l = _emulate_two_ranks(lambda r, w, ex: pipeline.evaluate_auc(
            enc, head, "LTN", dataset, world[feats], world[txt], world[masks], 3, 16, return_frames=True, rank=r, world=w, exchange=ex))
        assert np.max(np.abs(s - G[f"ev/{tag}/scores"])) < 2e-6 and np.array_equal(l, G[f"ev/{tag}/labels"])
        assert abs(auc - float(G[f"ev/{tag}/auc"][0])) < 1e-6
    a1 = pipeline.evaluate_train_auc(enc, head, "LTN", "SHT", world["sht_feats"], world["sht_train"], world["sht_masks"], 3, 16,
                                     return_frames=True)
    a2 = _emulate_two_ranks(lambda r, w, ex: pipeline.evaluate_train_auc(
        enc, head, "LTN", "SHT", world["sht_feats"], world["sht_train"], world["sht_masks"], 3, 16, return_frames=True, rank=r, world=w,
        exchange=ex))
    assert np.array_equal(a1[2], a2[2]) and np.max(np.abs(a1[1] - a2[1])) < 2e-6 and abs(a1[0] - a2[0]) < 1e-6
    # UCF: a non-owner reads only the list line and the ground truth of a video
    enc = _OracleEncoder(world["ltn_ucf_enc.ckpt"], pw.LTN_UCF); head = _OracleHead(world["ltn_ucf_cls.ckpt"], "classifier")
    auc, s, l = _emulate_two_ranks(lambda r, w, ex: pipeline.evaluate_auc(
        enc, head, "LTN", "UCF", world["ucf_feats"], world["ucf_test"], world["ucf_gt"], 2, 9, return_frames=True, rank=r, world=w,
        exchange=ex))
    assert np.max(np.abs(s - G["ev/ucf/scores"])) < 2e-6 and np.array_equal(l, G["ev/ucf/labels"])
    two = _emulate_two_ranks(lambda r, w, ex: pipeline.generate_pseudo_labels(
        enc, head, "LTN", "UCF", world["ucf_feats"], world["ucf_train"], 0.25, part_len=2, n_patch=9, rank=r, world=w, exchange=ex))
    _check_pseudo(two, "pl/t_ucf/", 0.25)
    enc = _OracleEncoder(world["stn_sht_enc.ckpt"], pw.STN_SHT); head = _OracleHead(world["stn_sht_reg.ckpt"], "regressor")
    two = _emulate_two_ranks(lambda r, w, ex: pipeline.generate_pseudo_labels(
        enc, head, "STN", "SHT", world["sht_feats"], world["sht_train"], 0.34, rank=r, world=w, exchange=ex))
    _check_pseudo(two, "pl/s_sht/", 0.34)
