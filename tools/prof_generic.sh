#!/bin/bash
# rocprofv3 passes over the teacher and KD-step legs (generic row-major path): kernel trace + two counter groups.
#   tools/prof_generic.sh <out_dir_under_gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for leg in teacher kd_step; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${leg}_trace" -- python3 $GRAFT_REPO_ROOT/tools/${leg}_profile.py > "$OUT/${leg}_trace.log" 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${leg}_pmc1" -- python3 $GRAFT_REPO_ROOT/tools/${leg}_profile.py > "$OUT/${leg}_pmc1.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/${leg}_pmc2" -- python3 $GRAFT_REPO_ROOT/tools/${leg}_profile.py > "$OUT/${leg}_pmc2.log" 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
grep -c . "$OUT/summary.txt"
