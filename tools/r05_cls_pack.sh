#!/bin/bash
# CLS-only layer on the activation stream's pack: tests, then the bf16 step and its kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_cls
O=gpurun_out/r05_cls
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/tests_act16.log 2>&1; echo "act16 tests exit $?" >> $O/tests_act16.log
tail -5 $O/tests_act16.log
for r in 1 2; do
  timeout 300 python bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $O/bench_bf16_$r.json 2> $O/bench_bf16.err
  LSTC_CLS_PACK=0 timeout 300 python bench.py --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $O/bench_bf16_clsf32_$r.json 2>> $O/bench_bf16.err
done
grep -o '"ms_per_step": [0-9.]*' $O/bench_bf16_*.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o bf16 -- python3 "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 > "$GRAFT_REPO_ROOT/$O/bench_prof.json" 2> "$GRAFT_REPO_ROOT/$O/bench_prof.err"
cd "$GRAFT_REPO_ROOT"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python tools/kernel_table.py "$f" > $O/kernel_table_bf16.md
head -40 $O/kernel_table_bf16.md
