#!/bin/bash
# rocprofv3 kernel table of one bench config / dtype: tools/profile_config.sh <tag> <config> <fp32|bf16>  -> gpurun_out/prof_<tag>/
TAG=$1; CFG=$2; DT=$3
R="$PWD"; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$R"
D=gpurun_out/rp_${CFG}_$DT; rm -rf $D
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 bench.py --config $CFG --dtype $DT --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 \
    > $OUT/${TAG}_bench_under_rocprof_${CFG}_$DT.json 2> /dev/null
python3 tools/summarize_rocprof.py stats $D $OUT/${TAG}_bench_${CFG}_kernel_stats_$DT.md "$CFG step, $DT mode, rocprofv3 --kernel-trace --stats -- python3 bench.py --config $CFG --dtype $DT --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 (10 steps incl. warm-up; fp32 runs add no event steps, 16-bit modes add 3)"
rm -rf $D
