#!/bin/bash
# Round 6: per-bucket optimizer steps (Adagrad + bf16 weight repack of bucket k on a side stream as soon as ITS
# all-reduce has landed) against the all-buckets-then-one-step order - the RCCL bucket path forced onto ONE rank (LSTC_FORCE_DIST=1:
# a real one-rank communicator; EMULATED rank of the 8-GPU split at 4 pairs per rank, and the full batch).  Same box, alternating.
#   tools/r06_bucket_steps_ab.sh <tag>
TAG=${1:-r06}
R="$PWD"; OUT=$R/gpurun_out/bucket_ab_$TAG; mkdir -p $OUT
export LSTC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
port=29560
for rep in 1 2; do
 for bs in 4 32; do
  for dt in bf16 fp32; do
   for b in 0 1; do
    port=$((port + 1))
    MASTER_PORT=$port LSTC_BUCKET_STEPS=$b timeout 600 python bench.py --config ltn_sht --batch_size $bs --dtype $dt --no-extras --no-cpu-baseline \
       --no-h2d --steps 30 --warmup 5 > $OUT/${TAG}_force_dist_bs${bs}_${dt}_bucketsteps${b}_rep$rep.json 2> $OUT/err_bs${bs}_${dt}_${b}_$rep.txt
   done
  done
 done
done
python3 - <<'PY' $OUT
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*_force_dist_*.json")):
    try:
        o = json.load(open(f)); c = o["config"]
        print(f.split("/")[-1], "ms/step", o["ms_per_step"], "median", o["ms_per_step_median"], "backward", c.get("backward_ms_per_step"), "exposed", c.get("comm_exposed_ms_per_step"), "loss", o["loss_last_timed_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
