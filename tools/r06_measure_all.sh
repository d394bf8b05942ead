#!/bin/bash
# Round 6: every measurement the profiles/ directory quotes for the final code, in one box run.   tools/r06_measure_all.sh
R="$PWD"; OUT=$R/gpurun_out/r06_final; mkdir -p $OUT
timeout 900 python bench.py > $OUT/r06_bench_default.json 2> $OUT/bench_default.err
bash tools/profile_round.sh r06 > $OUT/profile_round.log 2>&1
cp gpurun_out/prof_r06/* $OUT/ 2>/dev/null
bash tools/bench_fp32_others.sh r06 > $OUT/others_fp32.txt 2>&1
bash tools/bench_bf16_all.sh r06 > $OUT/others_bf16.txt 2>&1
cp gpurun_out/r06_bench_*.json $OUT/ 2>/dev/null
bash tools/bench_rank_shapes.sh r06 > $OUT/rank_shapes.txt 2>&1
cp gpurun_out/rank_r06/r06_rank_*.json $OUT/ 2>/dev/null
python3 tools/gemm_launch_table.py bf16 > $OUT/r06_gemm_launch_table_bf16.txt 2> /dev/null
python3 tools/gemm_launch_table.py fp32 > $OUT/r06_gemm_launch_table_fp32.txt 2> /dev/null
for dt in fp32 bf16; do
  timeout 300 python tools/eval_throughput.py $dt > $OUT/r06_eval_throughput_$dt.json 2> /dev/null
  timeout 600 python tools/coteach_round.py --dtype $dt > $OUT/r06_coteach_round_$dt.json 2> /dev/null
done
ls $OUT | head -80
