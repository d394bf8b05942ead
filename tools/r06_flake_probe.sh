#!/bin/bash
# Round 6: the mixed UBnormal + SHT step (BASELINE config 5) in fp32 on ONE rank and on EIGHT ranks sharing the box's one GPU over gloo,
# six times: the loss of the timed steps must be the same numbers every time (one run of 13 inside the full test suite was not:
# tests/test_dist_gpu.py::test_bench_eight_ranks_split_the_batch_of_one_rank).   tools/r06_flake_probe.sh
A="--steps 2 --warmup 1 --batch_size 32 --part_num 8 --no-extras --no-h2d --no-cpu-baseline --config mixed_ubn_sht --dtype fp32 --no-dropout --max_clips 200"
P='import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o["loss_first_timed_step"], o["loss_last_timed_step"])'
for i in 1 2 3 4 5 6; do
  a=$(python bench.py $A 2>/dev/null | python -c "$P")
  b=$(LSTC_SHARE_DEVICE=1 LSTC_DIST_BACKEND=gloo MASTER_PORT=$((29700+i)) python bench.py --gpus 8 $A 2>/dev/null | python -c "$P")
  echo "run $i: one rank [$a]   eight ranks [$b]"
done
