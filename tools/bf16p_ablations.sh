#!/bin/bash
# Timing ablations of the packed bf16 GEMM (tools/tuning build only: `make tuning`; LstcGemmDesc.variant >> 4 selects them and the
# products are WRONG by construction).  Bits: 1 = no epilogue (accumulators kept live), 4 = no steady-state LDS-DMA,
# 8 = no fragment reads, 16 = epilogue arithmetic without store instructions, 32 = every tile stores into rows 0-255 (output stays in L2),
# 64 = every DMA reads K step 0 of tile (0, 0) (same instruction stream, always cache hits), 128 = hand-counted epilogue without the
# quad transposes / 16-lane exchange (alone: the full kernel with values stored in the wrong places).
# Usage (on an MI355X):  tools/bf16p_ablations.sh > profiles/rNN_gemm_bf16p_ablations.log
G=tools/tuning/gemm_check
for shape in "100352 2048 2048 0 1" "100352 2048 6144 0 1" "2048 2048 100352 1 0"; do
  for abl in 0 128 1 16 32 5 65 9 13; do
    case $abl in 0) n="full kernel";; 128) n="full kernel, epilogue without transposes";; 1) n="no epilogue";; 16) n="epilogue arithmetic, no stores";; 32) n="stores stay in L2";;
      5) n="no epilogue, no DMA";; 65) n="no epilogue, DMA from a cache-hot tile";; 9) n="no epilogue, no fragment reads";; 13) n="MFMA only (no epilogue, DMA, fragment reads)";; esac
    split=1; [ "$shape" = "2048 2048 100352 1 0" ] && split=4
    echo -n "$n | "; timeout 60 $G one $shape $((abl * 16)) $split 0 20 0 0 3 | grep TIME
  done
done
