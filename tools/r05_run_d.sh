R="$PWD"; OUT=$R/gpurun_out/r05_d; mkdir -p $OUT
python tools/gemm_launch_table.py bf16 > $OUT/gemm_launch_table_bf16.txt 2>&1
for fl in 0 128 136 648 652 653; do
  ./tools/gemm_check one 100352 2048 2048 0 1 0 1 $fl 10 0 0 3 | grep TIME >> $OUT/gemm_epi_times.txt
done
for fl in 8 13 136 653; do
  ./tools/gemm_check one 100352 2048 4096 0 1 0 1 $fl 10 0 0 3 | grep TIME >> $OUT/gemm_epi_times.txt
done
./tools/gemm_check one 100352 2048 6144 0 1 0 1 648 10 0 0 3 | grep TIME >> $OUT/gemm_epi_times.txt
./tools/gemm_check one 100352 2048 6144 0 1 0 1 8 10 0 0 3 | grep TIME >> $OUT/gemm_epi_times.txt
cat $OUT/gemm_launch_table_bf16.txt | head -30; cat $OUT/gemm_epi_times.txt
