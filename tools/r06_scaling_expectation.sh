#!/bin/bash
# Round 6 (VERDICT r5 upkeep): the inputs of DESIGN 5's PRE-REGISTERED expectation for the driver's N = 1 / 2 / 4 / 8 scaling run - one rank
# of the strong-scaled headline job EMULATED on one GPU (global batch 32 pairs: 16 / 8 / 4 pairs per rank at N = 2 / 4 / 8; no
# communication, forced-RCCL bucket path so the step is the one an N-rank job runs), fp32 and bf16.  The model that turns these into
# expected step times is tools/scaling_model.py.      tools/r06_scaling_expectation.sh <tag>
TAG=${1:-r06}
R="$PWD"; OUT=$R/gpurun_out/scale_$TAG; mkdir -p $OUT
export LSTC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
port=29610
for dt in fp32 bf16; do
  for bs in 32 16 8 4; do
    port=$((port + 1))
    MASTER_PORT=$port timeout 600 python bench.py --config ltn_sht --batch_size $bs --dtype $dt --no-extras --no-cpu-baseline --no-h2d \
       --steps 20 --warmup 5 > $OUT/${TAG}_rank_ltn_sht_bs${bs}_${dt}.json 2> $OUT/err_${bs}_${dt}.txt
  done
done
python3 tools/scaling_model.py $OUT/${TAG}_rank_ltn_sht
