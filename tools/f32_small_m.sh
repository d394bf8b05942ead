#!/bin/bash
# exact-f32 GEMM at the per-rank shapes of an 8-GPU strong-scaled step (12544 tokens): one-tile-per-workgroup (variant 0) vs persistent walk (12)
for shape in "12544 2048 2048 0 1" "12544 2048 2048 0 0" "12544 4096 2048 0 1" "12544 2048 4096 0 1" "12544 6144 2048 0 1" "25088 2048 2048 0 1" "6272 2048 2048 0 1"; do
  for v in 0 12; do
    echo -n "shape $shape variant $v: "; ./tools/gemm_check one $shape $v 1 0 20 | tail -1
  done
done
