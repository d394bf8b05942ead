#!/bin/bash
# PMC passes (one counter group per run) over the CLS-only layer's passes on a packed X: tools/pmc_cls_pk.sh <out.txt>
OUT=${1:-gpurun_out/cls_pk_pmc.txt}
R="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$R"
: > "$OUT"
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=gpurun_out/pmc_tmp; rm -rf $D
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $D -o p -- python3 tools/cls_pk_time.py > /dev/null 2>&1 || echo "pass failed: $grp" >> "$OUT"
  python3 tools/summarize_rocprof.py pmc $D _pk_kernel 2>/dev/null | awk '{ if (!seen[$2 $3 $4]++) print }' | head -6 >> "$OUT"
done
rm -rf gpurun_out/pmc_tmp
cat "$OUT"
