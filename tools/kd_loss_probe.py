"""Diagnostic: time of the fused KD loss (forward + gradient) vs the same losses written with torch ops on the GPU."""
import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd.losses import CombinedKDLoss  # noqa: E402


def torch_combined(s, t, T=4.0, tau=0.05):
    mm = F.mse_loss(s - s.max(1, keepdim=True)[0], t / T - (t / T).max(1, keepdim=True)[0])
    lk = F.kl_div(F.log_softmax(s / T, 1), F.softmax(t / T, 1), reduction="batchmean") * T * T
    c = -F.log_softmax(s / tau, 1)[:, 0].mean()
    return 0.6 * mm + 0.2 * lk + 0.2 * c


fn = CombinedKDLoss()
for B in (64, 1024, 16384, 262144):
    s = torch.randn(B, 9, device="cuda", requires_grad=True)
    t = torch.randn(B, 9, device="cuda")
    out = {}
    for name, f in (("hip", lambda: fn(s, t)["loss"]), ("torch", lambda: torch_combined(s, t))):
        for _ in range(5):
            s.grad = None
            f().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            s.grad = None
            f().backward()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 50 * 1e6
    print(f"B={B} D=9: fused HIP {out['hip']:.0f} us, torch ops {out['torch']:.0f} us per forward+backward", flush=True)
