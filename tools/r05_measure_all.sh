#!/bin/bash
# Round 5: every measurement the profiles/ directory quotes, in one box run.   tools/r05_measure_all.sh
R="$PWD"; OUT=$R/gpurun_out/r05_final; mkdir -p $OUT
timeout 900 python bench.py > $OUT/r05_bench_default.json 2> $OUT/bench_default.err
bash tools/profile_round.sh r05 > $OUT/profile_round.log 2>&1
cp gpurun_out/prof_r05/* $OUT/ 2>/dev/null
bash tools/bench_fp32_others.sh r05 > $OUT/others_fp32.txt 2>&1
bash tools/bench_bf16_all.sh r05 > $OUT/others_bf16.txt 2>&1
cp gpurun_out/r05_bench_*.json $OUT/ 2>/dev/null
bash tools/bench_rank_shapes.sh r05 > $OUT/rank_shapes.txt 2>&1
cp gpurun_out/rank_r05/r05_rank_*.json $OUT/ 2>/dev/null
for ls in 1 1e-3; do
  timeout 300 python bench.py --lr_scale $ls --no-extras --no-cpu-baseline --no-h2d --steps 10 --warmup 3 > $OUT/r05_bench_lr_scale_$ls.json 2> /dev/null
done
for dt in fp32 bf16; do
  timeout 300 python tools/eval_throughput.py $dt > $OUT/r05_eval_throughput_$dt.json 2> /dev/null
  timeout 600 python tools/coteach_round.py --dtype $dt > $OUT/r05_coteach_round_$dt.json 2> /dev/null
done
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "^\[|passed|failed" > $OUT/r05_gpu_tests_printed.txt
ls $OUT | head -80; tail -3 $OUT/r05_gpu_tests_printed.txt
