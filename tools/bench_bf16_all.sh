#!/bin/bash
# bf16-mode bench lines of every config (one JSON line each) into gpurun_out/<prefix>_<config>_bf16.json
P=${1:-r03b}
mkdir -p gpurun_out
for c in ltn_sht stn_sht ltn_ucf ltn_ubnormal mixed_ubn_sht; do
  timeout 300 python bench.py --config $c --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${P}_bench_${c}_bf16.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${P}_bench_${c}_bf16.json"))
print("$c", d["ms_per_step"], d.get("ms_per_step_median"), d["value"])
PY
done
