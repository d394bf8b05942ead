for S in "100352 2048 2048 0 1" "100352 2048 6144 0 1"; do
echo -n "product     : "; tools/gemm_check one $S 0 1 0 20 0 0 3 | grep TIME | cut -c30-45,95-200
for dbg in 0 1 16 32 48; do
echo -n "tuning dbg=$dbg: "; build/tuning/gemm_check one $S $((16*dbg)) 1 0 20 0 0 3 | grep TIME | cut -c30-45,95-200
done
done
