#!/bin/bash
# Round 6: BASELINE config 4 (UCF) in fp32 on EIGHT ranks sharing the box's one GPU over gloo, N times each way: collectives on device
# tensors through torch's gloo staging (LSTC_GLOO_DEVICE_TENSORS=1: the path that returned wrong sums in about one run of six) against
# this repository's explicit host staging (lstc_vad_amd.dist.all_reduce_sum, the default under gloo).  The one-rank run of the same
# 64-video batch gives the reference numbers.       tools/r06_flake_probe8.sh [N]
N=${1:-12}
P='import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o["loss_first_timed_step"], o["loss_last_timed_step"])'
A="--config ltn_ucf --dtype fp32 --batch_size 32 --part_num 8 --no-dropout --steps 2 --warmup 1 --no-extras --no-h2d --no-cpu-baseline --max_clips 200"
ref=$(python bench.py $A 2>/dev/null | python -c "$P")
echo "one rank: $ref"
for mode in 1 0; do
  bad=0
  for i in $(seq 1 $N); do
    r=$(LSTC_GLOO_DEVICE_TENSORS=$mode LSTC_SHARE_DEVICE=1 LSTC_DIST_BACKEND=gloo MASTER_PORT=$((29900 + 20 * mode + i)) python bench.py --gpus 8 $A 2>/dev/null | python -c "$P")
    python3 - "$ref" "$r" <<'PY' || bad=$((bad + 1))
import sys
a = [float(x) for x in sys.argv[1].split()]; b = [float(x) for x in sys.argv[2].split()]
sys.exit(0 if all(abs(x - y) < 2e-5 for x, y in zip(a, b)) else 1)
PY
    echo "  eight ranks, $([ $mode = 1 ] && echo "torch's gloo staging of device tensors" || echo "explicit host staging"): $r"
  done
  echo "$([ $mode = 1 ] && echo "torch's gloo staging" || echo "explicit host staging"): $bad of $N runs disagree with the one-rank run"
done
