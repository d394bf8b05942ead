OUT=gpurun_out/attn_nt_ab; mkdir -p $OUT
for rep in 1 2 3; do
  for v in tree nt; do
    if [ $v = nt ]; then export LSTC_LIBRARY=$PWD/build/attn_nt/liblstc_hip.so; else unset LSTC_LIBRARY; fi
    timeout 300 python bench.py --config ltn_sht --dtype bf16 --no-extras --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $OUT/ab_${v}_$rep.json 2> /dev/null
    python3 -c "import json; o=json.load(open('$OUT/ab_${v}_$rep.json')); print('bf16 step attention DMA $v rep $rep: ms/step', o['ms_per_step'], 'median', o['ms_per_step_median'], 'gemm ms', o['roofline']['gemm_ms_per_step'], 'loss', o['loss_last_timed_step'])"
  done
done
