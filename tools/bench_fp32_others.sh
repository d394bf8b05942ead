#!/bin/bash
# fp32 bench lines of the non-headline configs into gpurun_out/<prefix>_bench_<config>.json
P=${1:-r03b}
mkdir -p gpurun_out
for c in stn_sht ltn_ucf ltn_ubnormal mixed_ubn_sht; do
  timeout 400 python bench.py --config $c --no-extras --no-cpu-baseline --no-h2d --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${P}_bench_${c}.json
  python - <<PY
import json
d = json.load(open("gpurun_out/${P}_bench_${c}.json"))
print("$c", d["ms_per_step"], d.get("ms_per_step_median"), d["value"], d["roofline"]["frac"])
PY
done
