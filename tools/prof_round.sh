#!/bin/bash
# Profile passes of a round (one gpurun call each part): kernel trace + stats, then separate --pmc passes.
#   [ROUND=r04] tools/prof_round.sh search|big|encode|generic      (output: gpurun_out/${ROUND}_prof)
set -e
ROUND=${ROUND:-r04}
LIB=$GRAFT_REPO_ROOT/semantic-search-kd_amd/libsskd_amd.so
OUT=$GRAFT_REPO_ROOT/gpurun_out/${ROUND}_prof; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # run <tag> <rocprof args...> : the program is tools/ab_search.py on the in-tree library
  local tag=$1; shift
  rocprofv3 "$@" --output-format csv -d "$OUT/$tag" -- python3 $GRAFT_REPO_ROOT/tools/ab_search.py "$LIB" > "$OUT/$tag.log" 2>&1 || echo "$tag failed" | tee -a "$OUT/errors.log"
  echo "$tag done"
}
case "$1" in
search)   # the bench shape: 10 000 queries x 1 M rows
  bash $GRAFT_REPO_ROOT/tools/prof_search.sh "$LIB" ${ROUND}_prof/screen_1m ;;
big)      # BASELINE cfg 3: 8 841 823 rows whole (screened and exact) and the 1 105 228-row shard of one of 8 ranks
  export AB_NOCHECK=1 AB_ROUNDS=3
  AB_ROWS=1105228 run shard_trace --kernel-trace --stats
  AB_ROWS=1105228 run shard_fetch --pmc FETCH_SIZE
  AB_ROWS=1105228 run shard_write --pmc WRITE_SIZE
  AB_ROWS=8841823 run whole_screened_trace --kernel-trace --stats
  AB_ROWS=8841823 run whole_screened_fetch --pmc FETCH_SIZE
  AB_ROWS=8841823 run whole_screened_write --pmc WRITE_SIZE
  AB_ROWS=8841823 AB_EXACT=1 run whole_exact_trace --kernel-trace --stats
  AB_ROWS=8841823 AB_EXACT=1 run whole_exact_fetch --pmc FETCH_SIZE
  AB_ROWS=8841823 AB_EXACT=1 run whole_exact_write --pmc WRITE_SIZE ;;
encode)
  bash $GRAFT_REPO_ROOT/tools/prof_encode.sh "$LIB" ${ROUND}_prof/encode ;;
generic)
  bash $GRAFT_REPO_ROOT/tools/prof_generic.sh ${ROUND}_prof/generic ;;
esac
