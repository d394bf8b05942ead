#!/bin/bash
# Round 6: does any kernel of the step read memory nobody wrote?  Every torch.empty() of the process is NaN / 0xFF-filled
# (LSTC_POISON_EMPTY=1 -> torch.utils.deterministic.fill_uninitialized_memory) and the losses of three steps are compared with the
# un-poisoned run - at the full batch and at ONE RANK's share of the 8-GPU split (4 pairs; forced one-rank RCCL bucket path), every
# config and GEMM mode.  A fresh process gets zero-filled device memory from the driver, a long-lived box does not: a read of
# uninitialised memory passes every isolated test and fails one run in a dozen inside a suite.      tools/uninit_probe.sh
P='import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o["loss_first_timed_step"], o["loss_last_timed_step"])'
port=29800
for cfg in ltn_sht ltn_ucf stn_sht ltn_ubnormal mixed_ubn_sht; do
  for dt in fp32 bf16 f32x3; do
    for bs in 4 32; do
      [ $bs = 32 ] && [ $cfg != ltn_sht ] && [ $cfg != mixed_ubn_sht ] && continue
      A="--config $cfg --dtype $dt --batch_size $bs --part_num 32 --no-dropout --steps 2 --warmup 1 --no-extras --no-h2d --no-cpu-baseline --max_clips 200"
      port=$((port + 2))
      a=$(LSTC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py $A 2>/dev/null | python -c "$P")
      b=$(LSTC_POISON_EMPTY=1 LSTC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((port + 1)) RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py $A 2>/dev/null | python -c "$P")
      [ "$a" = "$b" ] && v=same || v="DIFFERENT"
      echo "$cfg $dt pairs=$bs: plain [$a]  poisoned [$b]  $v"
    done
  done
done
