#!/bin/bash
# Round 5: the bf16 activation stream - parity tests, same-box A/B of the bf16 step (act bf16 vs fp32), kernel table.
R="$PWD"; OUT=$R/gpurun_out/r05_act; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$R"
timeout 900 python -m pytest tests/test_act16_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > $OUT/tests_act16.log
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "full_width_bf16 or packed_output_and_packed_relu or bf16_encoder_step" 2>&1 | tail -15 > $OUT/tests_bf16_existing.log
for i in 1 2; do
  for act in bf16 fp32; do
    timeout 300 python bench.py --dtype bf16 --act_dtype $act --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/bench_bf16_act_${act}_$i.json 2> $OUT/bench_bf16_act_${act}_$i.err
  done
done
D=gpurun_out/rp_act; rm -rf $D
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 \
      > $OUT/bench_under_rocprof_bf16.json 2> $OUT/rocprof_bf16.err
S=$(find $D -name "*kernel_stats.csv" | head -1)
if [ -n "$S" ]; then
  cp $S $OUT/kernel_stats_bf16.csv
  python3 tools/summarize_rocprof.py stats $D $OUT/kernel_stats_bf16.md "LTN-SHT step, bf16 mode + bf16 activation stream, rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 (10 steps incl. warm-up; round 5)"
fi
rm -rf $D
grep -h '"value"' $OUT/bench_bf16_act_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    o = json.loads(l); print(o['config'].get('act_dtype'), o['ms_per_step'], o['ms_per_step_median'], o['roofline']['achieved'], o['roofline']['gemm_ms_per_step'], o['hbm_peak_GB'])
"
tail -5 $OUT/tests_act16.log $OUT/tests_bf16_existing.log
