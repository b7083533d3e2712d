"""Diagnostic: single-query search latency through the host class (device-resident queries and host round trip)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd import FAISSIndexBuilder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
g = torch.Generator(device="cuda").manual_seed(1234)
c = torch.randn(n, 384, generator=g, device="cuda")
c = torch.nn.functional.normalize(c, dim=1)
ib = FAISSIndexBuilder(384, "Flat", "cosine")
ib.add(c)
del c
for nq, k in [(1, 10), (1, 100), (1, 200), (8, 10), (32, 10), (64, 10)]:
    q = torch.nn.functional.normalize(torch.randn(nq, 384, device="cuda"), dim=1)
    for _ in range(3):
        ib.search_device(q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ib.search_device(q, k)
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) / 20 * 1e3
    qh = q.cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(20):
        ib.search(qh, k)
    host_ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"n={n} nq={nq} k={k}: device-resident {dev_ms:.3f} ms, host numpy in/out {host_ms:.3f} ms", flush=True)
