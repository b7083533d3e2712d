#!/bin/bash
# Same-box A/B of the two-branch teacher scoring: SSKD_FORWARD_STREAMS = 2 (default) against 1; teacher leg of bench.py.
set -e
B="python bench.py --corpus 200000 --queries 2000 --steps 6 --warmup 2 --no-cpu-baseline --no-encode --no-train --no-hostile --no-cfg3"
for mode in 2 1 2 1; do
  export SSKD_FORWARD_STREAMS=$mode
  timeout -k 10 400 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['teacher']
print('streams $mode:', d['value'], 'pairs/s', d['ms_per_step'], 'ms', d['roofline']['frac'], 'text', (d.get('text') or {}).get('value'))"
done
