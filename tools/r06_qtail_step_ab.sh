#!/bin/bash
# Round 6: same-box A/B of the quarter-tile tail kernel at STEP level (bf16 mode): LSTC_P1_QTAIL=0 against the default, alternating,
# on the headline step (batch 32 pairs) and the per-rank shapes of 2 / 4 / 8 GPUs (16 / 8 / 4 pairs).   tools/r06_qtail_step_ab.sh
OUT=gpurun_out/qtail_ab; mkdir -p $OUT
for rep in 1 2 3; do
  for q in 0 2; do
    for bs in 32 4; do
      LSTC_P1_QTAIL=$q timeout 300 python bench.py --config ltn_sht --batch_size $bs --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 \
        > $OUT/ab_bs${bs}_q${q}_$rep.json 2> /dev/null
      python3 -c "import json; o=json.load(open('$OUT/ab_bs${bs}_q${q}_$rep.json')); print('ltn_sht bs $bs QTAIL=$q rep $rep: ms/step', o['ms_per_step'], 'median', o['ms_per_step_median'], 'gemm ms', o['roofline']['gemm_ms_per_step'], 'loss', o['loss_last_timed_step'])"
    done
  done
done
for q in 0 2; do
  for cfg in ltn_ucf mixed_ubn_sht; do
    LSTC_P1_QTAIL=$q timeout 300 python bench.py --config $cfg --batch_size 4 --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/ab_${cfg}_q$q.json 2> /dev/null
    python3 -c "import json; o=json.load(open('$OUT/ab_${cfg}_q$q.json')); print('$cfg bs 4 QTAIL=$q: ms/step', o['ms_per_step'], 'gemm ms', o['roofline']['gemm_ms_per_step'], 'loss', o['loss_last_timed_step'])"
  done
  for bs in 16 8; do
    LSTC_P1_QTAIL=$q timeout 300 python bench.py --config ltn_sht --batch_size $bs --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/ab_bs${bs}_q$q.json 2> /dev/null
    python3 -c "import json; o=json.load(open('$OUT/ab_bs${bs}_q$q.json')); print('ltn_sht bs $bs QTAIL=$q: ms/step', o['ms_per_step'], 'gemm ms', o['roofline']['gemm_ms_per_step'], 'loss', o['loss_last_timed_step'])"
  done
done
