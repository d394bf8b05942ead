R="$PWD"; OUT=$R/gpurun_out/r05_b; mkdir -p $OUT
./build/pack_access_probe > $OUT/pack_access_probe.txt 2>&1
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "headline_size_backward or full_width_training_step_matches" 2>&1 | grep -v "^$" | tail -40 > $OUT/tests_tight.log
timeout 900 python tools/coteach_round.py --dtype bf16 > $OUT/coteach_bf16.json 2> $OUT/coteach_bf16.err
timeout 1500 bash tools/bench_rank_shapes.sh r05 > $OUT/rank_shapes.txt 2>&1
cat $OUT/pack_access_probe.txt; tail -25 $OUT/tests_tight.log; cat $OUT/coteach_bf16.json | head -c 3000; tail -3 $OUT/coteach_bf16.err; cat $OUT/rank_shapes.txt
