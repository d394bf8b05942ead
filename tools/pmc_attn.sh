#!/bin/bash
# PMC passes (rocprofv3 --pmc, one counter group per run) over the attention kernels of one layer shape.
# Usage (on an MI355X): tools/pmc_attn.sh <out.txt> <attn_one.py args>
OUT=$1; shift
R="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$R"
: > "$OUT"
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  D=gpurun_out/pmc_tmp; rm -rf $D
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $D -o p -- python3 tools/attn_one.py "$@" > /dev/null 2>&1 || echo "pass failed: $grp" >> "$OUT"
  python3 tools/summarize_rocprof.py pmc $D attn_ 2>/dev/null | tail -2 >> "$OUT"
done
rm -rf gpurun_out/pmc_tmp
cat "$OUT"
