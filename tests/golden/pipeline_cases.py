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


# ---------------------------------------------------------------------------------------------- config-3 chain
# One set of command lines for the reference's Train/*.py parsers (golden generator) and for this repo's Train/*.py CLI
# (tests/test_pipeline_gpu.py): STN -> labels -> LTN -> labels -> STN co-teaching, on the SHT dialect of the world.
CHAIN = dict(seed=3, epochs=3, lr_encoder="1e-3", lr_head="2e-3")


def chain_argv(stage: str, P: dict):
    """``P``: world paths + save (dir), {stn,ltn}_{enc,reg,cls}_{in,out} checkpoints, pl_s / pl_t label files, pl_mce (no
    suffix), thr_s."""
    C = CHAIN
    common = ["--dataset_path", P["sht_feats"], "--training_txt", P["sht_train"], "--testing_txt", P["sht_test"],
              "--test_mask_dir", P["sht_masks"], "--model_save_dir", P["save"], "--batch_size", "2", "--part_num", "3",
              "--n_patch", "16", "--n_head", "2", "--d_model", "32", "--d_k", "16", "--d_v", "16", "--load_model",
              "--lr_encoder", C["lr_encoder"], "--inter_epoch", "1000", "--seed", str(C["seed"]), "--save_threshold", "2",
              "--position_dropout", "0"]
    if stage == "stn":
        return common + ["--epochs", str(C["epochs"]), "--part_len", "2", "--n_hidden", "47", "--FFN_layerNorm", "--train_dataset", P["sht_feats"],
                         "--load_spatio_model_path", P["stn_sht_enc.ckpt"], "--load_classifier_model_path", P["stn_sht_reg.ckpt"],
                         "--num_workers", "2", "--MHA_attn_dropout", "0", "--MHA_fc_dropout", "0", "--FFN_dropout", "0",
                         "--regressor_dropout", "0", "--saved_prefix", "", "--lr_regressor", C["lr_head"]]
    if stage == "ltn":
        return common + ["--epochs", str(C["epochs"]), "--part_len", "3", "--n_hidden", "64", "--FFN_layerNorm", "--MHA_layerNorm",
                         "--relative_position_encoding", "--load_temporal_model_path", P["ltn_enc_in"],
                         "--load_classifier_model_path", P["ltn_cls_in"], "--pseudo_labels_path", P["pl_s"],
                         "--MHA_attn_dropout", "0", "--MHA_fc_dropout", "0", "--FFN_dropout", "0", "--classifier_dropout", "0",
                         "--saved_prefix", "chain_", "--lr_classifier", C["lr_head"]]
    if stage == "mce":
        return common + ["--spatio_epochs", str(C["epochs"]), "--spatio_part_len", "2", "--spatio_n_hidden", "47", "--spatio_FFN_layerNorm",
                         "--spatio_model_path", P["stn_enc_out"], "--regression_model_path", P["stn_reg_out"],
                         "--spatio_pseudo_path", P["pl_t"], "--temporal_pseudo_path", P["pl_mce"], "--threshold", P["thr_s"],
                         "--num_workers", "2", "--spatio_MHA_attn_dropout", "0", "--spatio_MHA_fc_dropout", "0",
                         "--spatio_FFN_dropout", "0", "--regressor_dropout", "0", "--saved_prefix", "mce_", "--lr_regressor", C["lr_head"]]
    raise KeyError(stage)


def weight_fingerprint(state_dict, n=48):
    """(L2 norm per tensor, ``n`` strided samples per tensor) of a state_dict, in key order - float tensors only."""
    import numpy as np
    norms, samp = [], []
    for k, v in state_dict.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if a.dtype.kind != "f":
            continue
        a = a.astype(np.float64).ravel()
        norms.append(np.sqrt((a * a).sum()))
        idx = (np.arange(n, dtype=np.int64) * 2654435761 + 12345) % a.size
        samp.append(a[idx])
    return np.array(norms), np.concatenate(samp)
