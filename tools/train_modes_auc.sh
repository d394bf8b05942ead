#!/bin/bash
# Same synthetic LTN training run (learnable: abnormal videos carry a brighter stretch, lstc_vad_amd/data.py) in the three
# GEMM modes; prints the loss trajectory ends and the test AUCs the training loop logs.  BASELINE config 5's "bf16 + fp32
# AUC parity check" in miniature, plus the f32x3 mode.   usage: tools/train_modes_auc.sh [out_dir] [steps]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
O=${1:-$ROOT/gpurun_out/modes}; STEPS=${2:-60}; mkdir -p $O; O="$(cd "$O" && pwd)"
cd "$ROOT/Train"
for DT in fp32 f32x3 bf16; do
  python temporal_transformer_shanghaitech.py --synthetic --synthetic_pairs 32 --d_model 512 --n_head 8 --d_k 64 --d_v 64 \
    --n_hidden 1024 --n_patch 16 --part_len 3 --part_num 8 --batch_size 8 --MHA_layerNorm --FFN_layerNorm \
    --relative_position_encoding --encoder_weight_init --classifier_weight_init --epochs 1000 --inter_epoch 5 --steps $STEPS \
    --lr_classifier 1e-3 --compute_dtype $DT --log_dir $O/$DT > $O/$DT.out 2>&1
  echo "== $DT: first/last loss lines, AUC lines"
  grep "loss" $O/$DT.out | head -1 | cut -c25-; grep "loss" $O/$DT.out | tail -1 | cut -c25-
  grep "test AUC" $O/$DT.out | cut -c25- | tr '\n' ';'; echo
done
