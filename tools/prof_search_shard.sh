#!/bin/bash
# Kernel trace of the screened search on an 8-GPU-sized shard (AB_ROWS rows, default 125000):
#   tools/prof_search_shard.sh <lib.so> <out_dir_under_gpurun_out> [rows]
set -e
LIB=$(realpath "$1"); OUT=$GRAFT_REPO_ROOT/gpurun_out/$2; mkdir -p "$OUT"
export AB_ROWS=${3:-125000}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $GRAFT_REPO_ROOT/tools/ab_search.py "$LIB" > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/trace/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'at::' in n or 'rocclr' in n:
        continue
    print(n.replace('(anonymous namespace)::', '')[:60], r['Calls'], r['AverageNs'])
PY
