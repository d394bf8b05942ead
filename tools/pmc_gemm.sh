#!/bin/bash
# PMC passes (rocprofv3 --pmc, one counter group per run: MI355X_MICROARCH.md "rocprofv3 PMC slots") over single GEMM launches of
# tools/gemm_check.  Usage: tools/pmc_gemm.sh <out.txt> <key in kernel name> <gemm_check one ... args>
# Output lines: dispatch id, kernel, {counter: sum over the chip}.  FETCH_SIZE / WRITE_SIZE are KB (gfx950: x2 on FETCH_SIZE for
# wide reads); SQ_VALU_MFMA_BUSY_CYCLES counts per-SIMD cycles; effective clock = GRBM_GUI_ACTIVE / 8 / kernel time.
set -e
OUT=$1; KEY=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
: > "$OUT"
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  D=gpurun_out/pmc_tmp; rm -rf $D
  rocprofv3 --pmc $grp --output-format csv -d $D -o p -- ./tools/gemm_check one "$@" > /dev/null 2>&1 || echo "pass failed: $grp" >> "$OUT"
  python3 tools/summarize_rocprof.py pmc $D "$KEY" | tail -2 >> "$OUT"
done
rm -rf gpurun_out/pmc_tmp
cat "$OUT"
