"""Dataset-class parity cases shared by make_golden_pipeline.py (reference side) and tests/test_pipeline_host.py."""

_SH = dict(feats="sht_feats", txt="sht_train")
_UCF = dict(cls="UCF_Train_Origin_Dataset", feats="ucf_feats", txt="ucf_train", part_len=2, n_patch=9)
_UBN = dict(feats="ubn_feats", txt="ubn_train", n_patch=16)

DATASET_CASES = {
    "sh_uniform": dict(cls="SH_Train_Origin_Dataset", seed=1, part_num=3, part_len=2, n_patch=16, sample="uniform", **_SH),
    "sh_random_pseudo": dict(cls="SH_Train_Origin_Dataset", seed=2, part_num=2, part_len=3, n_patch=4, sample="random",
                             pseudo="sht_pseudo", **_SH),
    "sh_npatch1": dict(cls="SH_Train_Origin_Dataset", seed=3, part_num=4, part_len=1, n_patch=1, sample="uniform", **_SH),
    "sh_mutual_uniform_pseudo": dict(cls="SH_Train_Origin_Dataset_MutualTraining", seed=4, part_num=3, part_len=3, n_patch=16,
                                     sample="uniform", pseudo="sht_pseudo", **_SH),
    "sh_mutual_random": dict(cls="SH_Train_Origin_Dataset_MutualTraining", seed=5, part_num=5, part_len=1, n_patch=16,
                             sample="random", **_SH),
    "sh_tencrop_uniform": dict(cls="SH_Train_Origin_Dataset_tenCrop", seed=6, part_num=3, part_len=2, n_patch=4, d_model=8,
                               sample="uniform", pseudo="sht10_pseudo", feats="sht10_feats", txt="sht_train"),
    "sh_tencrop_random": dict(cls="SH_Train_Origin_Dataset_tenCrop", seed=7, part_num=2, part_len=3, n_patch=4, d_model=8,
                              sample="random", feats="sht10_feats", txt="sht_train"),
    "ucf_uniform": dict(seed=8, part_num=4, sample="uniform", **_UCF),
    "ucf_random_pseudo": dict(seed=9, part_num=3, sample="random", pseudo="ucf_pseudo", **_UCF),
    "ucf_crop_return": dict(cls="UCF_Train_Origin_Dataset", seed=10, part_num=3, part_len=2, n_patch=4, d_model=8,
                            sample="uniform", crop_return=True, feats="ucf10_feats", txt="ucf_train"),
    "ubn_uniform": dict(cls="UBnormal_Train_Origin_Dataset", seed=11, part_num=3, part_len=5, sample="uniform", **_UBN),
    "ubn_random": dict(cls="UBnormal_Train_Origin_Dataset", seed=12, part_num=2, part_len=2, sample="random", **_UBN),
    "ubn_tencrop": dict(cls="UBnormal_Train_Origin_Dataset_tenCrop", seed=13, part_num=2, part_len=2, n_patch=4, d_model=8,
                        sample="uniform", feats="ubn10_feats", txt="ubn_train"),
}


def build_dataset(mod, spec, W):
    """Construct ``mod.<cls>`` (the reference's ``utils.load_dataset`` or ``lstc_vad_amd.load_dataset``) for a case."""
    cls = getattr(mod, spec["cls"])
    pseudo = W[spec["pseudo"]] if spec.get("pseudo") else None
    a = (spec["part_num"], spec["part_len"])
    if spec["cls"].startswith("UCF"):
        return cls(*a, 16, W[spec["feats"]], W[spec["txt"]], spec["n_patch"], spec["sample"], pseudo_labels_path=pseudo,
                   d_model=spec.get("d_model", 4096), crop_return=spec.get("crop_return", False))
    if spec["cls"].endswith("tenCrop"):
        return cls(*a, W[spec["feats"]], W[spec["txt"]], spec["n_patch"], spec["sample"], spec["d_model"],
                   pseudo_labels_path=pseudo)
    return cls(*a, W[spec["feats"]], W[spec["txt"]], spec["n_patch"], spec["sample"], pseudo_labels_path=pseudo)
