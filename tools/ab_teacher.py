"""Same-box A/B of the teacher leg (and the KD step) between library builds: ``python tools/ab_teacher.py lib_a.so lib_b.so``
runs ``bench_support.bench_teacher`` / ``bench_kd_step`` in one child process per library (the package loads ONE library
per process), alternately, twice."""
import json
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CHILD = """
import json, sys, torch
from pathlib import Path
sys.path.insert(0, %r)
from semantic_search_kd_amd import _native
_native._LIB_PATH = Path(sys.argv[1]).resolve()
from semantic_search_kd_amd.bench_support import bench_teacher, bench_kd_step
dev = torch.device("cuda:0")
t = bench_teacher(dev, 1, 5, 2, torch.cuda.synchronize)
k = bench_kd_step(dev, steps=5, warmup=2)
print(json.dumps({"teacher_ms": t["ms_per_step"], "teacher_frac": t["roofline"]["frac"], "kd_ms": k["ms_per_step"]}))
""" % str(REPO)
for rnd in range(2):
    for lib in sys.argv[1:]:
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(Path(lib).stem, line[-1] if line else out.stderr[-400:], flush=True)
