#!/bin/bash
# Instruction-mix / stall PMC passes over one gemm_check launch (see tools/pmc_gemm.sh).  Usage: tools/pmc_gemm2.sh <out.txt> <key> <gemm_check one args>
OUT=$1; KEY=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
: > "$OUT"
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_CMD_FIFO_FULL" \
           "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE32_CYCLES SQ_IFETCH" \
           "SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_WAVE32_LDS SQ_INST_CYCLES_SALU"; do
  D=gpurun_out/pmc_tmp; rm -rf $D
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $D -o p -- ./tools/gemm_check one "$@" > /dev/null 2>&1 || echo "pass failed: $grp" >> "$OUT"
  python3 tools/summarize_rocprof.py pmc $D "$KEY" 2>/dev/null | tail -1 >> "$OUT"
done
rm -rf gpurun_out/pmc_tmp
cat "$OUT"
