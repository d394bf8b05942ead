R="$PWD"; OUT=$R/gpurun_out/r05_c; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_act16_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -30 > $OUT/tests_new.log
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "headline_size_backward or full_width_training_step_matches or bf16" > $OUT/tests_tight.log 2>&1
for i in 1 2; do
  for act in bf16 fp32; do
    timeout 300 python bench.py --dtype bf16 --act_dtype $act --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/bench_bf16_act_${act}_$i.json 2> $OUT/bench_bf16_act_${act}_$i.err
  done
done
LSTC_EAGER_GATHER=1 timeout 300 python bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/bench_bf16_eager_gather.json 2> /dev/null
timeout 300 python bench.py --dtype fp32 --no-extras --no-cpu-baseline --no-h2d --steps 10 --warmup 3 > $OUT/bench_fp32.json 2> /dev/null
D=gpurun_out/rp_act; rm -rf $D
cd /tmp && export TMPDIR=/tmp; cd "$R"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 \
      > $OUT/bench_under_rocprof_bf16.json 2> $OUT/rocprof_bf16.err
S=$(find $D -name "*kernel_stats.csv" | head -1)
if [ -n "$S" ]; then
  cp $S $OUT/kernel_stats_bf16.csv
  python3 tools/summarize_rocprof.py stats $D $OUT/kernel_stats_bf16.md "LTN-SHT step, bf16 mode + bf16 activation stream, rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 (10 steps incl. warm-up; round 5)"
fi
rm -rf $D
timeout 600 python tools/coteach_round.py --dtype bf16 > $OUT/coteach_bf16.json 2> $OUT/coteach_bf16.err
python3 - <<'PY' $OUT
import sys, json, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1]); r = o.get("roofline") or {}
        print(f.split("/")[-1], o["config"].get("act_dtype"), o["ms_per_step"], o["ms_per_step_median"], r.get("achieved"), r.get("gemm_ms_per_step"), o["hbm_peak_GB"], o["loss_last_timed_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -6 $OUT/tests_new.log; grep -c "un-aligned" $OUT/tests_tight.log; tail -3 $OUT/tests_tight.log
python3 -c "
import json,sys
o=json.load(open('$OUT/coteach_bf16.json'))
for k,v in o.items():
    if isinstance(v,dict) and ('ms_per_step' in v or 'clips_per_s' in v): print(k, {a:v[a] for a in ('ms_per_step','snippets_per_s','clips_per_s','wall_s','eval_s') if a in v})
print('round', o.get('round_wall_s'))"
