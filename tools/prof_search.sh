#!/bin/bash
# rocprofv3 passes over the screened search at the bench shape (tools/ab_search.py, one library):
#   tools/prof_search.sh <lib.so> <out_dir_under_gpurun_out>
set -e
LIB=$(realpath "$1"); OUT=$GRAFT_REPO_ROOT/gpurun_out/$2; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $GRAFT_REPO_ROOT/tools/ab_search.py "$LIB" > "$OUT/trace.log" 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python3 $GRAFT_REPO_ROOT/tools/ab_search.py "$LIB" > "$OUT/pmc$i.log" 2>&1 || echo "pmc group $i failed: $grp" | tee -a "$OUT/errors.log"
done
