#!/bin/bash
# Round 6: same-box A/B of the item walk of the persistent packed bf16 GEMM: the reference build (build/qtail: round-major walk, the code
# before the XCD-contiguous walk), and the in-tree library with LSTC_P1_XWALK = 0 (round-major) / 1 (XCD-contiguous chunks, equal split) /
# 2 (chunks split by the XCDs' measured speeds).  Launch level, then the bf16 step.   tools/r06_xwalk_ab.sh
run() {   # M N K tA tB variant split flags
  printf "ref      "; build/qtail/gemm_check one $1 $2 $3 $4 $5 $6 $7 $8 20 0 0 3 | grep TIME
  for x in 0 1 2; do
    printf "XWALK=%s  " $x
    LSTC_P1_XWALK=$x tools/gemm_check one $1 $2 $3 $4 $5 $6 $7 $8 20 0 0 3 | grep TIME
  done
}
run 100352 2048 2048 0 1 0 1 128
run 100352 2048 2048 0 1 0 1 652
run 100352 2048 4096 0 1 0 1 653
run 100352 2048 6144 0 1 0 1 648
run 100352 4096 2048 0 1 0 1 131
run 100352 4096 2048 0 1 0 1 400
run 100352 6144 2048 0 1 0 1 128
run 2048 2048 100352 1 0 7 4 0
run 6144 2048 100352 1 0 7 4 0
run 4096 2048 100352 1 0 7 2 0
run 50176 4096 2048 0 1 0 1 131
run 25088 6144 2048 0 1 0 1 128
OUT=gpurun_out/xwalk_ab; mkdir -p $OUT
for rep in 1 2 3; do
  for v in ref 0 1 2; do
    if [ $v = ref ]; then export LSTC_LIBRARY=$PWD/build/qtail/liblstc_hip.so; unset LSTC_P1_XWALK; else unset LSTC_LIBRARY; export LSTC_P1_XWALK=$v; fi
    for bs in 32 8; do
      timeout 300 python bench.py --config ltn_sht --batch_size $bs --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/ab_bs${bs}_${v}_$rep.json 2> /dev/null
      python3 -c "import json; o=json.load(open('$OUT/ab_bs${bs}_${v}_$rep.json')); print('ltn_sht bs $bs walk $v rep $rep: ms/step', o['ms_per_step'], 'median', o['ms_per_step_median'], 'gemm ms', o['roofline']['gemm_ms_per_step'], 'loss', o['loss_last_timed_step'])"
    done
  done
done
