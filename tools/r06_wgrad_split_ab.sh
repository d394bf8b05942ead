#!/bin/bash
# Round 6: same-box A/B of the split-K choice of the packed bf16 weight gradients: LSTC_WGRAD_COST_MODEL=0 (round filling alone, rounds 2 - 5)
# against 1 (functional._wgrad_split_bf16p: K-loop time + per-item fixed cost + the ordered sum of the partials).   tools/r06_wgrad_split_ab.sh
OUT=gpurun_out/wgrad_ab; mkdir -p $OUT
for rep in 1 2; do
  for cfgbs in "ltn_sht 32" "ltn_sht 4" "ltn_ucf 4" "stn_sht 4" "stn_sht 32" "mixed_ubn_sht 4"; do
    set -- $cfgbs
    for m in 0 1; do
      LSTC_WGRAD_COST_MODEL=$m timeout 300 python bench.py --config $1 --batch_size $2 --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/ab_$1_$2_$m_$rep.json 2> /dev/null
      python3 -c "import json; o=json.load(open('$OUT/ab_$1_$2_$m_$rep.json')); print('$1 bs $2 cost_model=$m rep $rep: ms/step', o['ms_per_step'], 'median', o['ms_per_step_median'], 'gemm ms', o['roofline']['gemm_ms_per_step'], 'loss', o['loss_last_timed_step'])"
    done
  done
done
