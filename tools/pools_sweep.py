"""Diagnostic: batch search time vs number of queries, with the scan's pruning pools forced on / off."""
import subprocess
import sys
import time
from pathlib import Path

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from semantic_search_kd_amd import FAISSIndexBuilder, _native

    n = int(sys.argv[2])
    c = torch.nn.functional.normalize(torch.randn(n, 384, device="cuda"), dim=1)
    ib = FAISSIndexBuilder(384, "Flat", "cosine")
    ib.add(c)
    ib.search_tuning = _native.SearchTuning(0, 0, 1 if sys.argv[3] == "1" else -1)
    for nq in [64, 128, 256, 512, 1024, 2048, 4096]:
        q = torch.nn.functional.normalize(torch.randn(nq, 384, device="cuda"), dim=1)
        for _ in range(2):
            ib.search_device(q, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ib.search_device(q, 10)
        torch.cuda.synchronize()
        print(f"  nq={nq}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)
else:
    for n in (1_000_000, 125_000):
        for pools in ("1", "0"):
            print(f"n={n} pools={pools}", flush=True)
            subprocess.run([sys.executable, __file__, "child", str(n), pools], check=True)
