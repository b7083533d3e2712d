"""Is the KD step launch-bound?  Host time to ENQUEUE a step vs time until the GPU has finished it, plus a
cProfile of the host side: ``python tools/kd_host_probe.py``."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd.bench_support import synthetic_ids  # noqa: E402
from semantic_search_kd_amd.losses import CombinedKDLoss  # noqa: E402
from semantic_search_kd_amd.training import TrainableEncoder  # noqa: E402
from semantic_search_kd_amd.weights import BertConfig, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = BertConfig()
model = TrainableEncoder(cfg, synthetic_state_dict(cfg), dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-5)
loss_fn = CombinedKDLoss()
q_ids, q_mask = synthetic_ids(32, 32, cfg.vocab_size, dev, seed=1)
d_ids, d_mask = synthetic_ids(256, 256, cfg.vocab_size, dev, seed=2)
teacher = torch.randn((32, 8), device=dev) * 3.0


def step():
    opt.zero_grad(set_to_none=True)
    q = model(q_ids, q_mask)
    d = model(d_ids, d_mask).view(32, 8, -1)
    out = loss_fn(torch.einsum("th,tdh->td", q, d), teacher)
    out["loss"].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 5:.1f} ms/step, until done {1e3 * (t2 - t0) / 5:.1f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
