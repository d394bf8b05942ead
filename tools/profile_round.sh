#!/bin/bash
# One round's measurement evidence, on an MI355X box (gpurun): rocprofv3 kernel tables of the bench step in fp32 and bf16 mode,
# the bench lines printed by those same runs, PMC passes over the dominant GEMM launches and the attention kernels.
# Usage: tools/profile_round.sh <tag, e.g. r03>      -> gpurun_out/prof_<tag>/  (copy what is to be judged into profiles/)
TAG=${1:-r03}
R="$PWD"; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$R"
for dt in fp32 bf16; do
  D=gpurun_out/rp_$dt; rm -rf $D
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 bench.py --dtype $dt --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 \
      > $OUT/${TAG}_bench_under_rocprof_$dt.json 2> $OUT/rocprof_$dt.err
  S=$(find $D -name "*kernel_stats.csv" | head -1)
  if [ -n "$S" ]; then
    cp $S $OUT/${TAG}_bench_ltn_sht_kernel_stats_$dt.csv
    python3 tools/summarize_rocprof.py stats $D $OUT/${TAG}_bench_ltn_sht_kernel_stats_$dt.md "LTN-SHT step, $dt mode, rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype $dt --no-extras --no-cpu-baseline --no-h2d --steps 7 --warmup 3 (10 steps incl. warm-up; round ${TAG#r})"
  fi
  rm -rf $D
done
# PMC: exact-f32 GEMM, NT 100352 x 2048 x 2048 (variant 0 = PIPE 5), and the packed bf16 GEMM forward / weight-gradient forms
GRAFT_REPO_ROOT=$R bash tools/pmc_gemm.sh $OUT/${TAG}_gemm_f32_pmc.txt gemm_f32 100352 2048 2048 0 1 0 1 0 3 > /dev/null 2>&1
GRAFT_REPO_ROOT=$R bash tools/pmc_gemm.sh $OUT/${TAG}_gemm_bf16p_pmc.txt gemm_bf16p 100352 2048 2048 0 1 0 1 0 3 0 0 3 > /dev/null 2>&1
GRAFT_REPO_ROOT=$R bash tools/pmc_gemm.sh $OUT/${TAG}_gemm_bf16p_tr_pmc.txt gemm_bf16p 2048 2048 100352 1 0 0 4 0 3 0 0 3 > /dev/null 2>&1
# round 5: the same NT launch as the bf16 activation stream runs it - packed output + packed residual (flags 8 + 128 + 512)
GRAFT_REPO_ROOT=$R bash tools/pmc_gemm.sh $OUT/${TAG}_gemm_bf16p_act_pmc.txt gemm_bf16p 100352 2048 2048 0 1 0 1 648 3 0 0 3 > /dev/null 2>&1
# round 6: the launch WITHOUT a per-element operand (packed output only, flags 128): where the 16-byte epilogue shows
GRAFT_REPO_ROOT=$R bash tools/pmc_gemm.sh $OUT/${TAG}_gemm_bf16p_out_pmc.txt gemm_bf16p 100352 2048 2048 0 1 0 1 128 3 0 0 3 > /dev/null 2>&1
bash tools/pmc_attn.sh $OUT/${TAG}_attn_pmc_fp32.txt fp32 49 3 2048 256 > /dev/null 2>&1
bash tools/pmc_attn.sh $OUT/${TAG}_attn_pmc_bf16.txt bf16 49 3 2048 256 packed > /dev/null 2>&1
ls -la $OUT
