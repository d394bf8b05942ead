#!/bin/bash
# Round 6: same-box A/B of the packed-output epilogue of csrc/gemm_bf16p.hip - build/r05 (round 5: 32 x 8-B stores per wave behind quad
# transposes), build/epi16a (16 x 16-B stores on operand-swapped accumulators, operand wait in front of each unit), in-tree (the same with
# the wait behind the unit's own arithmetic) - on the bf16 headline step and the STN / UBnormal steps, alternating, 3 rounds.
OUT=gpurun_out/epi16_ab; mkdir -p $OUT
for rep in 1 2 3; do
  for v in r05 epi16a tree; do
    lib=$PWD/build/$v/liblstc_hip.so; [ $v = tree ] && lib=$PWD/lstc_vad_amd/liblstc_hip.so
    for cfg in ltn_sht; do
      LSTC_LIBRARY=$lib timeout 300 python bench.py --config $cfg --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 \
        > $OUT/ab_${cfg}_${v}_$rep.json 2> /dev/null
      python3 -c "import json; o=json.load(open('$OUT/ab_${cfg}_${v}_$rep.json')); print('$cfg $v rep $rep: ms/step', o['ms_per_step'], 'median', o['ms_per_step_median'], 'GEMM TFLOP/s', o['roofline']['achieved'], 'gemm ms', o['roofline']['gemm_ms_per_step'], 'loss', o['loss_last_timed_step'])"
    done
  done
done
for v in r05 tree; do
  lib=$PWD/build/$v/liblstc_hip.so; [ $v = tree ] && lib=$PWD/lstc_vad_amd/liblstc_hip.so
  for cfg in stn_sht ltn_ubnormal ltn_ucf; do
    LSTC_LIBRARY=$lib timeout 300 python bench.py --config $cfg --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/ab_${cfg}_${v}.json 2> /dev/null
    python3 -c "import json; o=json.load(open('$OUT/ab_${cfg}_${v}.json')); print('$cfg $v: ms/step', o['ms_per_step'], 'GEMM TFLOP/s', o['roofline']['achieved'])"
  done
done
