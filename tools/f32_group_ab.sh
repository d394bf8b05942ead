#!/bin/bash
# A/B of the exact-f32 GEMM's tile order inside an XCD (build/<name> from tools/build_variant.sh <name> gemm_f32 -DLSTC_F32_GROUP_M=8 against
# the in-tree library): parity of the variant, timings of the step's shapes, and the L2-miss traffic (FETCH_SIZE) of the dominant shape.
V=${1:-grp8}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
echo "== check $V"; timeout 600 build/$V/gemm_check check 2>&1 | tail -1
for rep in 1 2; do
for v in base $V; do
  if [ $v = base ]; then G=tools/gemm_check; else G=build/$v/gemm_check; fi
  for shape in "100352 2048 2048 0 1 0 1 0" "100352 2048 2048 0 0 0 1 0" "100352 4096 2048 0 1 0 1 3" "100352 2048 4096 0 1 0 1 13" "100352 2048 4096 0 0 0 1 8" "100352 4096 2048 0 0 0 1 16" "12544 2048 2048 0 1 0 1 0"; do
    echo -n "$v: "; timeout 120 $G one $shape 10 | grep TIME
  done
done
done
for v in base $V; do
  if [ $v = base ]; then G=./tools/gemm_check; else G=./build/$v/gemm_check; fi
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    D=gpurun_out/pmc_tmp; rm -rf $D
    rocprofv3 --pmc $grp --output-format csv -d $D -o p -- $G one 100352 2048 2048 0 1 0 1 0 3 > /dev/null 2>&1
    echo -n "$v $grp: "; python3 tools/summarize_rocprof.py pmc $D gemm_f32 | tail -1
  done
done
rm -rf gpurun_out/pmc_tmp
