"""Diagnostic: a few single-query searches for rocprofv3 --kernel-trace --stats."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd import FAISSIndexBuilder  # noqa: E402

n = 1_000_000
c = torch.nn.functional.normalize(torch.randn(n, 384, device="cuda"), dim=1)
ib = FAISSIndexBuilder(384, "Flat", "cosine")
ib.add(c)
k = int(sys.argv[1])
q = torch.nn.functional.normalize(torch.randn(1, 384, device="cuda"), dim=1)
for _ in range(5):
    ib.search_device(q, k)
torch.cuda.synchronize()
