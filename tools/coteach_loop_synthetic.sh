#!/bin/bash
# BASELINE.json config 3 in miniature: STN train -> STN pseudo labels -> LTN train on them -> LTN pseudo labels ->
# STN co-teaching (MIL + BCE) on those.  Four CLI invocations chained by .npy files, like README.md:21-36 upstream.
set -e
cd "$(dirname "$0")/../Train"
O=${1:-../gpurun_out/coteach}; mkdir -p $O
DT=${2:-fp32}        # fp32 | f32x3 | bf16 (BASELINE config 3 is quoted in bf16)
ST=${3:-6}           # optimisation steps per training stage
M="--d_model 128 --d_k 16 --d_v 16 --n_patch 16 --synthetic --synthetic_pairs 8 --compute_dtype $DT"
python spatio_transformer_shanghaitech.py $M --n_hidden 203 --FFN_layerNorm --encoder_weight_init --regressor_weight_init --batch_size 4 --part_num 4 --part_len 2 --steps $ST --inter_epoch 100 --log_dir $O/stn 2>&1 | tail -1
python pseudo_labels_generator_spatio.py $M --n_hidden 203 --FFN_layerNorm --threshold 0.5 --pseudo_labels_path $O/STN_pseudo_labels.npy 2>&1 | tail -1
python temporal_transformer_shanghaitech.py $M --n_hidden 256 --part_len 3 --MHA_layerNorm --FFN_layerNorm --relative_position_encoding --pseudo_labels_path $O/STN_pseudo_labels.npy --batch_size 4 --part_num 4 --steps $ST --inter_epoch 100 --log_dir $O/ltn 2>&1 | tail -1
python pseudo_labels_generator_temporal.py $M --n_hidden 256 --part_len 3 --MHA_layerNorm --FFN_layerNorm --relative_position_encoding --threshold 0.4 --pseudo_labels_path $O/LTN_pseudo_labels.npy 2>&1 | tail -1
python spatio_transformer_MIL_CE.py $M --spatio_n_hidden 203 --spatio_FFN_layerNorm --spatio_part_len 2 --spatio_pseudo_path $O/LTN_pseudo_labels.npy --batch_size 4 --part_num 4 --steps $ST --inter_epoch 100 --log_dir $O/mce 2>&1 | tail -1
python - <<PY
import numpy as np
for f in ("$O/STN_pseudo_labels.npy", "$O/LTN_pseudo_labels.npy"):
    d = np.load(f, allow_pickle=True).tolist()
    k = sorted(d)[0]
    print(f, len(d), "videos;", k, d[k].shape, "nonzero frac %.2f" % float((np.concatenate([v.ravel() for v in d.values()]) > 0).mean()))
PY
