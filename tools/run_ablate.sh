#!/bin/bash
# A/B of build/<name> variants against the in-tree library (tools/build_variant.sh): correctness of each variant, then timings
for v in "$@"; do echo "== check $v"; timeout 300 build/$v/gemm_check check 2>&1 | tail -1; done
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then G=tools/gemm_check; else G=build/$v/gemm_check; fi
  for shape in "100352 2048 2048 0 1 0 1 0" "100352 4096 2048 0 1 0 1 3" "100352 2048 4096 0 1 0 1 12" "100352 2048 6144 0 1 0 1 8" "2048 2048 100352 1 0 0 4 0" "6144 2048 100352 1 0 0 4 0"; do
    echo -n "$v: "; timeout 60 $G one $shape 20 0 0 3 | grep TIME | sed -e "s/TIME bf16p//" -e "s/var=.*: //"
  done
done
done
