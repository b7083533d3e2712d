"""Turn the rocprofv3 passes of ``tools/prof_round.sh search|big`` into ``profiles/<round>/search_traffic.json`` and
``profiles/screen_traffic.json`` (what ``bench.py`` reads for ``roofline.traffic``):
``python tools/make_search_traffic.py gpurun_out/<round>_prof``.

Per search CALL: the screening step is ONE launch of ``screen_append_kernel`` (its bound-only sample phase runs inside
it; builds that still launch a separate pre-pass - template argument ``true`` - are summed per call).
FETCH_SIZE is doubled (gfx950 tallies 128-byte requests as 64 bytes), WRITE_SIZE is used as read; both are in KB."""
import csv
import os
import re
import hashlib
import json
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
ROUND = os.environ.get("ROUND", "r04")
root = Path(sys.argv[1])
NOTE = ("separate rocprofv3 passes (tools/prof_round.sh): --kernel-trace --stats for durations, --pmc FETCH_SIZE and --pmc "
        "WRITE_SIZE alone for traffic; FETCH_SIZE doubled per the gfx950 rule (128-B requests tallied as 64 B), WRITE_SIZE as "
        "read; HBM bytes = (2 FETCH + WRITE) x 1024 per search call (one launch of the screening kernel); program "
        "tools/ab_search.py (10 000 queries, k = 10, seeded unit rows)")


def is_prepass(name: str) -> bool:
    return re.search(r"screen_append_kernel<\d+,\d+,\d+,true", name.replace(" ", "")) is not None


def newest(d: Path, pattern: str):
    files = sorted(d.rglob(pattern), key=lambda f: f.stat().st_mtime)
    return files[-1] if files else None


def durations(d: Path, key: str):
    """(calls, ms per call avg, ms per call min) of the kernels whose name contains ``key``; a call = one main launch"""
    f = newest(d, "*kernel_trace.csv")
    rows = [r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls, cur = [], 0.0
    for r in rows:
        cur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if not is_prepass(r["Kernel_Name"]):   # the bound-only pre-pass precedes its main pass
            calls.append(cur)
            cur = 0.0
    calls = calls[1:] if len(calls) > 1 else calls              # first call: cold caches / code load
    return len(calls), sum(calls) / len(calls), min(calls)


def counter(d: Path, key: str, name: str):
    f = newest(d, "*counter_collection.csv")
    tot, mains = 0.0, 0
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"] and r["Counter_Name"] == name:
            tot += float(r["Counter_Value"])
            if not is_prepass(r["Kernel_Name"]):
                mains += 1
    return tot / mains


def entry(trace: Path, fetch: Path, write: Path, key: str, rows: int, flops: float):
    n, avg, mn = durations(trace, key)
    f, w = counter(fetch, key, "FETCH_SIZE"), counter(write, key, "WRITE_SIZE")
    hbm = (2.0 * f + w) * 1024.0
    return {"kernel": key, "rows": rows, "queries": 10000, "launches_profiled": n, "kernel_ms_avg": round(avg, 3),
            "kernel_ms_min": round(mn, 3), "fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1),
            "hbm_bytes_per_launch": hbm, "hbm_counter_gbs": round(hbm / (avg * 1e-3) / 1e9, 1),
            "tflops": round(flops / (avg * 1e-3) / 1e12, 1)}


sha = hashlib.sha256((REPO / "semantic-search-kd_amd" / "csrc" / "search.hip").read_bytes()).hexdigest()[:16]
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()
out = {"meta": {"note": NOTE, "search_hip_sha": sha, "from": f"tools/make_search_traffic.py@{head}"}}
fl = lambda rows: 2.0 * 384 * rows * 10000
s1 = root / "screen_1m"
if s1.exists():
    out["screen_1m"] = entry(s1 / "trace", s1 / "pmc4", s1 / "pmc5", "screen_append_kernel", 1000000, fl(1000000))
for tag, name, key, rows in (("shard", "screen_shard_1105228", "screen_append_kernel", 1105228),
                             ("whole_screened", "screen_8841823", "screen_append_kernel", 8841823),
                             ("whole_exact", "exact_8841823", "scan_topk_kernel", 8841823)):
    if (root / f"{tag}_trace").exists():
        out[name] = entry(root / f"{tag}_trace", root / f"{tag}_fetch", root / f"{tag}_write", key, rows, fl(rows))
if "exact_8841823" in out:   # B_q = 64 queries per pass at this shape
    passes = -(-10000 // 64)
    out["exact_8841823"]["algorithmic_hbm_bytes"] = passes * 8841823 * 1536
dst = REPO / "profiles" / ROUND / "search_traffic.json"
dst.write_text(json.dumps(out, indent=1))
print("wrote", dst)
if "screen_1m" in out:
    e = out["screen_1m"]
    (REPO / "profiles" / "screen_traffic.json").write_text(json.dumps({
        "kernel": "screen_append_kernel<10, 5, 12, 3, 2> (sample phase + slice phase in one launch)", "hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
        "fetch_size_kb": e["fetch_size_kb"], "write_size_kb": e["write_size_kb"], "search_hip_sha": sha, "note": NOTE,
        "from": f"profiles/{ROUND}/search_traffic.json@{head}"}, indent=1))
    print("wrote profiles/screen_traffic.json")
print(json.dumps({k: {kk: v[kk] for kk in ("kernel_ms_avg", "hbm_counter_gbs", "tflops")} for k, v in out.items() if k != "meta"}, indent=1))
