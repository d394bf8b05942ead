timeout 300 tools/gemm_check check 2>&1 | tail -1
for S in "100352 2048 2048 0 1 0 1 0" "100352 2048 2048 0 1 0 1 8" "100352 4096 2048 0 1 0 1 3" "100352 2048 6144 0 1 0 1 8"; do
echo -n "prev: "; build/prev/gemm_check one $S 20 0 0 3 | grep TIME | cut -c30-45,62-72,75-90,95-200
echo -n "now : "; tools/gemm_check one $S 20 0 0 3 | grep TIME | cut -c30-45,62-72,75-90,95-200
done
