R="$PWD"; OUT=$R/gpurun_out/r05_e; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_act16_gpu.py -q -m gpu -x -s 2>&1 | tail -30 > $OUT/tests_act16.log
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_width_gpu.py -q -m gpu -x -k "bf16" 2>&1 | tail -8 > $OUT/tests_bf16.log
for act in bf16 fp32; do
  timeout 300 python bench.py --config stn_sht --dtype bf16 --act_dtype $act --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/bench_stn_bf16_act_$act.json 2> $OUT/bench_stn_$act.err
  timeout 300 python bench.py --config ltn_ubnormal --dtype bf16 --act_dtype $act --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/bench_ubn_bf16_act_$act.json 2> /dev/null
done
timeout 300 python bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/bench_ltn_bf16.json 2> /dev/null
LSTC_FORCE_DIST=1 timeout 300 python bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 10 --warmup 3 > $OUT/bench_ltn_bf16_force_dist.json 2> $OUT/force_dist.err
timeout 600 python tools/coteach_round.py --dtype bf16 > $OUT/coteach_bf16.json 2> $OUT/coteach_bf16.err
python3 - <<'PY' $OUT
import sys, json, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1]); r = o.get("roofline") or {}
        print(f.split("/")[-1], o["config"].get("act_dtype"), o["ms_per_step"], o["ms_per_step_median"], r.get("achieved"), r.get("gemm_ms_per_step"), o["hbm_peak_GB"], o["loss_last_timed_step"], o["config"].get("comm_exposed_ms_per_step"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -6 $OUT/tests_act16.log; tail -3 $OUT/tests_bf16.log
python3 -c "
import json,sys
o=json.load(open('$OUT/coteach_bf16.json'))
for k,v in o.items():
    if isinstance(v,dict) and ('ms_per_step' in v or 'clips_per_s' in v): print(k, {a:v[a] for a in ('ms_per_step','snippets_per_s','clips_per_s','wall_s','eval_s') if a in v})
print('round', o.get('round_wall_s'), o.get('act_dtype'))"
