#!/bin/bash
# PMC passes over the packed-input attention kernels: tools/pmc_attn3.sh <out.txt> <attn_one.py args>
OUT=$1; shift
R="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$R"
: > "$OUT"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" "FETCH_SIZE" "WRITE_SIZE"; do
  D=gpurun_out/pmc_tmp; rm -rf $D
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $D -o p -- python3 tools/attn_one.py "$@" > /dev/null 2>&1 || echo "pass failed: $grp" >> "$OUT"
  python3 tools/summarize_rocprof.py pmc $D attn_ 2>/dev/null | tail -2 >> "$OUT"
done
rm -rf gpurun_out/pmc_tmp
cat "$OUT"
