"""A small synthetic "dataset world" on disk: feature archives, list files in the three dialects of the reference's
data/*.txt, frame masks, checkpoints.  Built from the portable generator (``lstc_vad_amd.synthetic``), so the golden
generator (reference side, this container) and the tests (build side, also on the GPU box) materialise byte-identical
files without shipping them.  Nothing here comes from the reference; formats follow SURVEY.md Appendix B.
"""
import os

import numpy as np
import torch

from lstc_vad_amd import synthetic as syn
from lstc_vad_amd.archive import write_archive

SEG = 16

# reduced-width models (same families as tests/golden/cases.py)
LTN_SHT = dict(n_layers=3, n_head=2, d_k=16, d_v=16, d_model=32, d_inner=64, MHA_layerNorm=True, FFN_layerNorm=True,
               relative_pe=True, window_size=4, window_depth=3)
STN_SHT = dict(n_layers=3, n_head=2, d_k=16, d_v=16, d_model=32, d_inner=47, FFN_layerNorm=True)
LTN_UCF = dict(n_layers=3, n_head=2, d_k=16, d_v=16, d_model=32, d_inner=64, MHA_layerNorm=True, FFN_layerNorm=True,
               relative_pe=True, window_size=4, window_depth=2)

SHT_TRAIN = [("01_0014", 1, 7), ("01_0016", 1, 30), ("01_002", 0, 12), ("02_001", 0, 3), ("03_004", 0, 26),
             ("04_0011", 1, 5), ("05_021", 0, 9), ("06_0147", 1, 17), ("07_005", 0, 21)]
SHT_TEST = [("01_001", 0, 11), ("01_0015", 1, 14), ("01_0025", 1, 2), ("08_003", 0, 6), ("09_0057", 1, 9)]
UCF_TRAIN = [("Vandalism/Vandalism048_x264", 23), ("Training_Normal_Videos_Anomaly/Normal_Videos826_x264", 40),
             ("Robbery/Robbery003_x264", 2), ("Training_Normal_Videos_Anomaly/Normal_Videos101_x264", 33),
             ("Abuse/Abuse007_x264", 70), ("Training_Normal_Videos_Anomaly/Normal_Videos002_x264", 1)]
UCF_TEST = [("Arson/Arson022_x264", 45, "Arson"), ("Testing_Normal_Videos_Anomaly/Normal_Videos_872_x264", 20, "Normal"),
            ("RoadAccidents/RoadAccidents021_x264", 37, "RoadAccidents")]
UBN_TRAIN = [("abnormal_scene_27_scenario_8", 13), ("normal_scene_3_scenario_1", 8), ("abnormal_scene_13_scenario_6", 10),
             ("normal_scene_9_scenario_2", 16), ("normal_scene_1_scenario_5", 6)]
UBN_TEST = [("abnormal_scene_1_scenario_1", 8), ("normal_scene_2_scenario_3", 5), ("abnormal_scene_1_scenario_4", 12)]


def fill_params(module: torch.nn.Module, seed: int):
    """Portable-generator weights, keyed by position in ``named_parameters()`` (identical order in the reference's and
    the build's classes)."""
    with torch.no_grad():
        for i, (k, p) in enumerate(module.named_parameters()):
            if k.endswith("layer_norm.weight"):
                v = 1.0 + syn.small_uniform(p.shape, seed, 100 + i, 0.2)
            elif k.endswith("bias") and p.dim() == 1:
                v = syn.small_uniform(p.shape, seed, 100 + i, 0.1)
            elif k.endswith("relative_position_bias_table"):
                v = syn.small_uniform(p.shape, seed, 100 + i, 0.5)
            else:
                v = syn.xavier_uniform(p.shape, seed, 100 + i)
            p.copy_(torch.from_numpy(v))


def _feats(n, P, d, stream):
    return syn.features((n, P, d), 11, stream).astype(np.float32)


def _mask(n_frames, stream):
    u = syn.uniform((n_frames,), 12, stream)
    m = np.zeros(n_frames, np.float64)
    a = int(u[0] * n_frames * 0.6)
    m[a:a + max(int(u[1] * n_frames * 0.4), 8)] = 1.0
    return m


def build(root, Encoder, Regressor, Classifier):
    """Materialise the world under ``root`` with the given model classes (the build's or the reference's - only used to
    enumerate parameters).  Returns a dict of paths."""
    os.makedirs(root, exist_ok=True)
    W = {"root": root}
    # ---- SHT dialect: P = 16, d = 32 (the generators feed whole arrays to the rel-PE LTN, so P must equal ws^2)
    arrays, stream = {}, 0
    for name, _, n in SHT_TRAIN + SHT_TEST:
        arrays[name + ".npy"] = _feats(n, 16, 32, stream); stream += 1
    W["sht_feats"] = write_archive(os.path.join(root, "sht_feats.npz"), arrays)
    W["sht_feats_dir"] = write_archive(os.path.join(root, "sht_feats_dir"), arrays)
    W["sht_train"] = os.path.join(root, "sht_train.txt")
    open(W["sht_train"], "w").write("".join(f"{n},{l}\n" for n, l, _ in SHT_TRAIN))
    W["sht_test"] = os.path.join(root, "sht_test.txt")
    open(W["sht_test"], "w").write("".join(f"{n},{l},{-1 if l else c * SEG + 3}\n" for n, l, c in SHT_TEST))
    W["sht_masks"] = os.path.join(root, "sht_masks") + os.sep
    os.makedirs(W["sht_masks"], exist_ok=True)
    for i, (n, l, c) in enumerate(SHT_TEST + SHT_TRAIN):
        if l:
            np.save(os.path.join(W["sht_masks"], n + ".npy"), _mask(c * SEG + 5, 50 + i))
    # ---- ten-crop SHT: [n*10*P, d] rows, P = 4, d = 8
    arrays = {}
    for i, (name, _, n) in enumerate(SHT_TRAIN):
        arrays[name + ".npy"] = _feats(n * 10, 4, 8, 200 + i).reshape(-1, 8)
    W["sht10_feats"] = write_archive(os.path.join(root, "sht10_feats.npz"), arrays)
    arrays = {p.split("/")[-1] + ".npy": _feats(n * 10, 4, 8, 230 + i).reshape(-1, 8) for i, (p, n) in enumerate(UCF_TRAIN)}
    W["ucf10_feats"] = write_archive(os.path.join(root, "ucf10_feats.npz"), arrays)
    arrays = {n + ".npy": _feats(c * 10, 4, 8, 260 + i).reshape(-1, 8) for i, (n, c) in enumerate(UBN_TRAIN)}
    W["ubn10_feats"] = write_archive(os.path.join(root, "ubn10_feats.npz"), arrays)
    # ---- UCF dialect: P = 9, d = 32; n_frames = clips*16 + a few
    arrays, gt = {}, {}
    for i, (path, n) in enumerate(UCF_TRAIN):
        arrays[path.split("/")[-1] + ".npy"] = _feats(n, 9, 32, 300 + i)
    for i, (path, n, lab) in enumerate(UCF_TEST):
        key = path.split("/")[-1]
        arrays[key + ".npy"] = _feats(n, 9, 32, 350 + i)
        if lab != "Normal":
            gt[key + ".npy"] = _mask(n * SEG + 7, 360 + i)
    W["ucf_feats"] = write_archive(os.path.join(root, "ucf_feats.npz"), arrays)
    W["ucf_gt"] = write_archive(os.path.join(root, "ucf_gt.npz"), gt)
    W["ucf_train"] = os.path.join(root, "ucf_train.txt")
    open(W["ucf_train"], "w").write("".join(f"{p}.mp4 {n * SEG + 7} \n" for p, n in UCF_TRAIN))
    W["ucf_test"] = os.path.join(root, "ucf_test.txt")
    open(W["ucf_test"], "w").write("".join(f"{p}.mp4 {n * SEG + 7} {lab} -1 -1 -1 -1 \n" for p, n, lab in UCF_TEST))
    # ---- UBnormal dialect: P = 16, d = 32 (scored with the SHT-width LTN)
    arrays = {}
    for i, (name, n) in enumerate(UBN_TRAIN + UBN_TEST):
        arrays[name + ".npy"] = _feats(n, 16, 32, 400 + i)
    W["ubn_feats"] = write_archive(os.path.join(root, "ubn_feats.npz"), arrays)
    W["ubn_train"] = os.path.join(root, "ubn_train.txt")
    open(W["ubn_train"], "w").write("".join(f"{n},{c * SEG + 3}\n" for n, c in UBN_TRAIN))
    W["ubn_test"] = os.path.join(root, "ubn_test.txt")
    open(W["ubn_test"], "w").write("".join(f"{n},{c * SEG + 3}\n" for n, c in UBN_TEST))
    W["ubn_masks"] = os.path.join(root, "ubn_masks") + os.sep
    os.makedirs(W["ubn_masks"], exist_ok=True)
    for i, (n, c) in enumerate(UBN_TEST):
        if n.startswith("abnormal"):
            np.save(os.path.join(W["ubn_masks"], n + ".npy"), _mask(c * SEG + 3, 450 + i))
    # ---- pseudo-label files in the generators' format (pickled dict, [n, 1] float32; one 2-column entry for :66-67)
    pl = {}
    for i, (name, _, n) in enumerate(SHT_TRAIN):
        pl[name + ".npy"] = syn.pseudo_labels((n, 1), 13, 0.65, i).astype(np.float32)
    pl["01_0016.npy"] = np.concatenate([1 - pl["01_0016.npy"], pl["01_0016.npy"]], axis=1)
    W["sht_pseudo"] = os.path.join(root, "sht_pseudo.npy")
    np.save(W["sht_pseudo"], pl)
    W["sht10_pseudo"] = os.path.join(root, "sht10_pseudo.npy")            # ten-crop variant keys have no ".npy"
    np.save(W["sht10_pseudo"], {k[:-4]: v for k, v in pl.items()})
    plu = {p.split("/")[-1] + ".npy": syn.pseudo_labels((max(n, 3) * 2, 1), 13, 0.65, 40 + i).astype(np.float32)
           for i, (p, n) in enumerate(UCF_TRAIN)}
    W["ucf_pseudo"] = os.path.join(root, "ucf_pseudo.npy")
    np.save(W["ucf_pseudo"], plu)
    # ---- checkpoints (state_dict files; the LTN-SHT pair carries the DataParallel "module." prefix)
    def ckpt(name, module, seed, prefix=""):
        fill_params(module, seed)
        path = os.path.join(root, name)
        torch.save({prefix + k: v.clone() for k, v in module.state_dict().items()}, path)
        W[name] = path
    ckpt("ltn_sht_enc.ckpt", Encoder(**LTN_SHT), 21, "module.")
    ckpt("ltn_sht_cls.ckpt", Classifier(32), 22, "module.")
    ckpt("stn_sht_enc.ckpt", Encoder(**STN_SHT), 23)
    ckpt("stn_sht_reg.ckpt", Regressor(32), 24)
    ckpt("ltn_ucf_enc.ckpt", Encoder(**LTN_UCF), 25)
    ckpt("ltn_ucf_cls.ckpt", Classifier(32), 26)
    return W
