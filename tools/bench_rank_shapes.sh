#!/bin/bash
# Per-rank shapes of the 8-GPU configs, EMULATED on ONE GPU (no communication: nothing here is a multi-GPU measurement):
#   headline LTN-SHT at 4 pairs per rank, config 4 (UCF: 4 + 4 videos x 32 parts x S = 19 = 4864 tokens per rank) and config 5
#   (UBnormal + SHT mixed: 2 + 2 pairs per model per rank), fp32 and bf16.   tools/bench_rank_shapes.sh <tag>
TAG=${1:-r05}
R="$PWD"; OUT=$R/gpurun_out/rank_$TAG; mkdir -p $OUT
for dt in fp32 bf16; do
  for cfg in ltn_sht ltn_ucf mixed_ubn_sht; do
    timeout 600 python bench.py --config $cfg --batch_size 4 --dtype $dt --no-extras --no-cpu-baseline --no-h2d --steps 30 --warmup 5 \
      > $OUT/${TAG}_rank_${cfg}_bs4_$dt.json 2> $OUT/rank_${cfg}_$dt.err
  done
done
python3 - <<'PY' $OUT
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*_rank_*.json")):
    try:
        o = json.load(open(f)); r = o.get("roofline") or {}
        print(f.split("/")[-1], o["ms_per_step"], o["value"], r.get("achieved"), r.get("gemm_ms_per_step"))
    except Exception as e:
        print(f, "unreadable", e)
PY
