#!/bin/bash
# Round 6: same-box A/B of the quarter-tile tail kernel of the packed bf16 GEMM (csrc/gemm_bf16p.hip, gemm_bf16p_q_kernel):
# LSTC_P1_QTAIL=0 (one persistent launch of 256 x 256 tiles) against 2 (default) and 3 rounds of quarter items.   tools/r06_qtail_ab.sh
make tools/gemm_check > /dev/null 2>&1
run() {   # M N K flags
  for q in 0 2 3; do
    printf "QTAIL=%s  " $q
    LSTC_P1_QTAIL=$q tools/gemm_check one $1 $2 $3 0 1 0 1 $4 20 0 0 3 | grep TIME
  done
}
echo "== headline shapes (100352 token rows)"
run 100352 2048 2048 128
run 100352 2048 2048 652
run 100352 2048 4096 653
run 100352 2048 6144 648
run 100352 4096 2048 131
run 100352 6144 2048 128
echo "== rank shapes (2 / 4 / 8 GPUs: 50176 / 25088 / 12544 rows)"
for m in 50176 25088 12544; do
  run $m 2048 2048 128
  run $m 2048 4096 653
  run $m 4096 2048 131
  run $m 6144 2048 128
done
echo "== the CLS-only layer's small products (f32 output)"
run 2048 2048 2048 0
run 2048 2048 4096 13
run 2048 512 2048 7
run 2048 4096 2048 131
