"""Same-box A/B of encoder variants: ``python tools/ab_encode.py [--rounds R] lib_a.so lib_b.so ...``
loads every library in ONE process and times the 512 x 256 forward alternately (A, B, A, B ...)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from semantic_search_kd_amd import _native  # noqa: E402
from semantic_search_kd_amd.bench_support import synthetic_ids  # noqa: E402
from semantic_search_kd_amd.weights import BertConfig, DeviceWeights, synthetic_state_dict  # noqa: E402

args = sys.argv[1:]
rounds, B, S = 6, 512, 256
while args and args[0].startswith("--"):
    k, v = args[0], args[1]
    args = args[2:]
    if k == "--rounds":
        rounds = int(v)
    elif k == "--batch":
        B = int(v)
    elif k == "--seq":
        S = int(v)
dev = torch.device("cuda:0")
cfg = BertConfig()
w = DeviceWeights(cfg, synthetic_state_dict(cfg), dev)
ids, mask = synthetic_ids(B, S, cfg.vocab_size, dev)
out = torch.empty((B, 384), dtype=torch.float32, device=dev)
libs = []
for path in args:
    lib = C.CDLL(str(Path(path).resolve()))
    for name in ("sskd_encoder_workspace_bytes", "sskd_encoder_forward"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _native.SIGNATURES[name]
    libs.append((Path(path).stem, lib))
ws = torch.empty(max(int(l.sskd_encoder_workspace_bytes(w.cstruct_cfg, B, S)) for _, l in libs), dtype=torch.uint8, device=dev)
st = int(torch.cuda.current_stream(dev).cuda_stream)


def run(lib, n):
    for _ in range(n):
        rc = lib.sskd_encoder_forward(w.cstruct_cfg, w.struct, ids.data_ptr(), mask.data_ptr(), B, S, 1,
                                      out.data_ptr(), ws.data_ptr(), ws.numel(), st)
        assert rc == 0, rc


ref = None
times = {n: [] for n, _ in libs}
for n, lib in libs:
    run(lib, 3)
    torch.cuda.synchronize()
    e = out.cpu().numpy().copy()
    if ref is None:
        ref = e
    print(f"{n}: max |diff| vs first variant {np.abs(e - ref).max():.3e}", flush=True)
for r in range(rounds):
    for n, lib in libs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(lib, 10)
        torch.cuda.synchronize()
        times[n].append((time.perf_counter() - t0) / 10 * 1e3)
for n, t in times.items():
    print(f"{n}: median {np.median(t):.4f} ms  min {np.min(t):.4f}  all {[round(x, 3) for x in t]}", flush=True)
