#!/bin/bash
# A/B of the packed bf16 GEMM's item order (build/<name> from tools/build_variant.sh <name> gemm_bf16p -DP1_SPLIT_MAJOR=1 -DP1_GROUP_M=4 ...)
# against the in-tree library: parity of each variant, timings of the step's shapes, L2-miss traffic of the weight-gradient form.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in "$@"; do echo "== check $v"; timeout 600 build/$v/gemm_check check 2>&1 | tail -1; done
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then G=tools/gemm_check; else G=build/$v/gemm_check; fi
  for shape in "100352 2048 2048 0 1 0 1 0" "100352 4096 2048 0 1 0 1 3" "100352 6144 2048 0 1 0 1 0" "100352 2048 4096 0 1 0 1 12" "100352 2048 6144 0 1 0 1 8" \
               "2048 2048 100352 1 0 0 4 0" "6144 2048 100352 1 0 0 4 0" "4096 2048 100352 1 0 0 2 0" "2048 4096 100352 1 0 0 2 0" "12544 2048 2048 0 1 0 1 0" "2048 2048 12544 1 0 0 4 0"; do
    echo -n "$v: "; timeout 60 $G one $shape 20 0 0 3 | grep TIME | sed -e "s/TIME bf16p//" -e "s/var=.*: //"
  done
done
done
for v in base "$@"; do
  if [ $v = base ]; then G=./tools/gemm_check; else G=./build/$v/gemm_check; fi
  for shape in "2048 2048 100352 1 0 0 4 0" "100352 6144 2048 0 1 0 1 0"; do
    D=gpurun_out/pmc_tmp; rm -rf $D
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D -o p -- $G one $shape 3 0 0 3 > /dev/null 2>&1
    echo -n "$v [$shape] FETCH_SIZE: "; python3 tools/summarize_rocprof.py pmc $D gemm_bf16p | tail -1 | cut -c1-20,110-
  done
done
rm -rf gpurun_out/pmc_tmp
