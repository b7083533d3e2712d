#!/bin/bash
# After `tools/prof_round.sh search` and `tools/prof_round.sh big` ran on a GPU box and gpurun merged their output:
# rebuild profiles/$ROUND/search_traffic.json, profiles/screen_traffic.json, the kernel-stats CSVs and the counter summary.
set -e
cd "$(dirname "$0")/.."
ROUND=${ROUND:-r04}
export ROUND
mkdir -p profiles/$ROUND
P=gpurun_out/${ROUND}_prof
python tools/make_search_traffic.py $P
cp "$(ls -t $P/screen_1m/trace/*/*kernel_stats.csv | head -1)" profiles/$ROUND/screen_1m_kernel_stats.csv
cp "$(ls -t $P/shard_trace/*/*kernel_stats.csv | head -1)" profiles/$ROUND/screen_shard_1105228_kernel_stats.csv
cp "$(ls -t $P/whole_screened_trace/*/*kernel_stats.csv | head -1)" profiles/$ROUND/screen_8841823_kernel_stats.csv
cp "$(ls -t $P/whole_exact_trace/*/*kernel_stats.csv | head -1)" profiles/$ROUND/exact_8841823_kernel_stats.csv
T=$(mktemp -d)
for d in screen_1m shard_trace shard_fetch shard_write whole_screened_trace whole_screened_fetch whole_screened_write whole_exact_trace whole_exact_fetch whole_exact_write; do cp -r $P/$d $T/; done
PMC_KEYS=screen_append,screen_finalize,scan_topk_kernel python tools/pmc_summary.py $T > profiles/$ROUND/search_profiles_summary.txt
rm -rf $T
