#!/bin/bash
# Shrink a rocprofv3 output tree before gpurun merges it back (limit: 64 MiB per call): write the per-kernel counter
# summary (tools/pmc_summary.py) next to it, keep the *kernel_stats.csv files, drop the raw traces and counter dumps.
#   tools/prof_pack.sh <dir under gpurun_out> [PMC_KEYS]
set -e
D=$GRAFT_REPO_ROOT/gpurun_out/$1
[ -n "$2" ] && export PMC_KEYS="$2"
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$D" > "$D/summary.txt" 2>/dev/null || true
find "$D" -type f \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" -o -name "*.db" -o -name "*domain_stats.csv" \) -delete
du -sh "$D" | cut -f1
