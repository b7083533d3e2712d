#!/bin/bash
# Same-box A/B of the split encoder forward: SSKD_FORWARD_STREAMS = 1 (one stream), 2 (default), 4; encode leg of bench.py.
set -e
B="python bench.py --corpus 200000 --queries 2000 --steps 10 --warmup 3 --no-cpu-baseline --no-ragged --no-text --no-train --no-teacher --no-hostile --no-cfg3"
for i in 1 2; do
  for mode in ${AB_MODES:-2 1 4}; do
    export SSKD_FORWARD_STREAMS=$mode
    timeout -k 10 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['encode']
print('streams $mode:', d['value'], 'docs/s', d['ms_per_step'], 'ms (graph)', d['ms_per_step_eager'], 'ms (eager)')"
  done
done
