#!/bin/bash
# Round 6: where a persistent workgroup of the packed bf16 GEMM spends its time, item by item (s_memtime stamps of the tuning build,
# csrc/gemm_bf16p.hip P1_STAMP): wait for K step 0 | K loop | next item's head DMA + epilogue | hand-over, for the step's K = 2048 /
# 4096 / 6144 products with the epilogues the bf16 activation stream uses, and for the epilogue-free / DMA-free ablations.
#   make tuning && tools/r06_bf16p_stamps.sh > profiles/r06_gemm_bf16p_stamps.txt
G=tools/tuning/gemm_check
S=$((0x2000))
run() { echo "== $1"; shift; timeout 120 $G one "$@" | grep -E "TIME|STAMPS"; }
# flags: 128 = OUT_PACK; 652 = OUT_PACK | RESIDUAL_PACK | RESIDUAL | DROPOUT; 131 = OUT_PACK | BIAS | RELU; 0 = f32 output
run "N=2048 K=2048 packed output (flags 128)"            100352 2048 2048 0 1 $((S * 16)) 1 128 20 0 0 3
run "N=2048 K=2048 packed residual + output (flags 652)" 100352 2048 2048 0 1 $((S * 16)) 1 652 20 0 0 3
run "N=2048 K=2048 packed residual + output, no dropout (flags 648)" 100352 2048 2048 0 1 $((S * 16)) 1 648 20 0 0 3
run "N=2048 K=4096 packed residual + output (flags 652)" 100352 2048 4096 0 1 $((S * 16)) 1 652 20 0 0 3
run "N=2048 K=6144 packed residual + output (flags 652)" 100352 2048 6144 0 1 $((S * 16)) 1 652 20 0 0 3
run "N=4096 K=2048 bias + ReLU, packed output (flags 131)" 100352 4096 2048 0 1 $((S * 16)) 1 131 20 0 0 3
run "N=2048 K=2048 f32 output (flags 0)"                 100352 2048 2048 0 1 $((S * 16)) 1 0 20 0 0 3
run "N=2048 K=2048 no epilogue"                          100352 2048 2048 0 1 $(((S + 1) * 16)) 1 0 20 0 0 3
run "N=2048 K=2048 no epilogue, DMA from a cache-hot tile" 100352 2048 2048 0 1 $(((S + 65) * 16)) 1 0 20 0 0 3
run "N=2048 K=2048 no epilogue, no DMA"                  100352 2048 2048 0 1 $(((S + 5) * 16)) 1 0 20 0 0 3
run "N=2048 K=6144 no epilogue"                          100352 2048 6144 0 1 $(((S + 1) * 16)) 1 0 20 0 0 3
run "rank shape 12544 x 2048 x 2048 (flags 128)"         12544 2048 2048 0 1 $((S * 16)) 1 128 20 0 0 3
run "weight gradient 2048 x 2048 x 100352, split 4"      2048 2048 100352 1 0 $((S * 16 + 7)) 4 0 20 0 0 3
# (XCD-staggered starts - workgroups of XCD x sleeping 3.2 / 6.4 us x x before their first item, so that the eight XCDs' epilogues do
# not coincide - were measured with a throw-away tuning bit on this script's shapes: profiles/r06_gemm_bf16p_stamps_stagger.txt.  No
# effect on the epilogue length (9.2 us with and without) or on the launch (0.752 vs 0.744 / 0.745 ms, 0.816 vs 0.815): the epilogue is
# not an HBM burst.  The code was removed again.)
