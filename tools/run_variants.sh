# A/B: in-tree library vs build/<name> variants on the bf16p GEMM shapes of the LTN step
timeout 300 tools/gemm_check check 2>&1 | tail -3
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then G=tools/gemm_check; else G=build/$v/gemm_check; fi
  for shape in "100352 2048 2048 0 1 0 1 0" "100352 4096 2048 0 1 0 1 3" "100352 2048 4096 0 1 0 1 12" "100352 2048 6144 0 1 0 1 8"; do
    echo -n "$v: "; timeout 60 $G one $shape 20 0 0 3 | grep TIME | cut -c1-12,95-200
  done
done
done
