"""Summarise rocprofv3 CSV output: per-kernel mean duration (kernel_trace) and per-kernel mean counter values
(counter_collection).  ``python tools/pmc_summary.py <dir>``; ``PMC_KEYS=a,b`` keeps only the kernels whose name
contains one of the keys."""
import csv
import os
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
KEYS = tuple(k for k in os.environ.get("PMC_KEYS", "mlp,attention,gemm_n384,scan_topk,screen_,gemm_nt").split(",") if k)
for f in sorted(root.rglob("*kernel_trace.csv")):
    d = defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"== {f.relative_to(root)}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if os.environ.get("PMC_KEYS") and not any(x in k for x in KEYS):
            continue
        print(f"  {k:60s} n={len(v):5d} mean={sum(v) / len(v):9.2f} us  total={sum(v) / 1e3:9.2f} ms")
for f in sorted(root.rglob("*counter_collection.csv")):
    d = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {f.relative_to(root)}")
    for k, cs in d.items():
        if not any(x in k for x in KEYS):
            continue
        print("  " + k)
        for c, v in cs.items():
            print(f"      {c:36s} n={len(v):5d} mean={sum(v) / len(v):16.1f}")
