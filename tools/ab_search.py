"""Same-box A/B of screened-search variants: ``python tools/ab_search.py lib_a.so lib_b.so ...`` loads every
library in ONE process and times the screening kernel (HIP events around it) alternately at the bench shape."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from semantic_search_kd_amd import _native  # noqa: E402

N, NQ, K = int(os.environ.get('AB_ROWS', 1_000_000)), int(os.environ.get('AB_NQ', 10_000)), 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
corpus = torch.nn.functional.normalize(torch.randn((N, 384), generator=g, device=dev), dim=1)
queries = torch.nn.functional.normalize(torch.randn((NQ, 384), generator=g, device=dev), dim=1)
hip = C.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
hip.hipEventSynchronize.argtypes = [C.c_void_p]
st = int(torch.cuda.current_stream(dev).cuda_stream)
names = ("sskd_index_tiled_bytes", "sskd_index_add_rows", "sskd_index_bf16_bytes", "sskd_index_make_bf16",
         "sskd_index_search_screened_workspace_bytes", "sskd_index_search_screened", "sskd_index_search_workspace_bytes",
         "sskd_index_search_profiled")
EXACT = bool(os.environ.get("AB_EXACT"))   # time the plain exact fp32 scan instead of the screened search
ROUNDS = int(os.environ.get("AB_ROUNDS", 7))
libs, ref = [], None
for path in sys.argv[1:]:
    lib = C.CDLL(str(Path(path).resolve()))
    for n in names:
        fn = getattr(lib, n)
        fn.restype, fn.argtypes = _native.SIGNATURES[n]
    tiled = torch.empty(int(lib.sskd_index_tiled_bytes(N)) // 4, dtype=torch.float32, device=dev)
    assert lib.sskd_index_add_rows(corpus.data_ptr(), N, 0, tiled.data_ptr(), 0, st) == 0
    bf = torch.empty(int(lib.sskd_index_bf16_bytes(N)), dtype=torch.uint8, device=dev)
    assert lib.sskd_index_make_bf16(tiled.data_ptr(), N, bf.data_ptr(), st) == 0
    ws = torch.empty(int(lib.sskd_index_search_workspace_bytes(N, NQ, K) if EXACT else
                         lib.sskd_index_search_screened_workspace_bytes(N, NQ, K)), dtype=torch.uint8, device=dev)
    libs.append((Path(path).stem, lib, tiled, bf, ws))
out_s = torch.empty((NQ, K), device=dev)
out_i = torch.empty((NQ, K), dtype=torch.int64, device=dev)
status = torch.zeros(2, dtype=torch.int32, device=dev)
times = {n: [] for n, *_ in libs}
wall = {n: [] for n, *_ in libs}
fallbacks = {n: 0 for n, *_ in libs}
for r in range(ROUNDS):
    for n, lib, tiled, bf, ws in libs:
        a, b = C.c_void_p(), C.c_void_p()
        hip.hipEventCreate(C.byref(a)); hip.hipEventCreate(C.byref(b))
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        if EXACT:
            rc = lib.sskd_index_search_profiled(tiled.data_ptr(), N, queries.data_ptr(), NQ, K, 0, out_s.data_ptr(),
                                                out_i.data_ptr(), ws.data_ptr(), ws.numel(), st, a, b)
        else:
            rc = lib.sskd_index_search_screened(tiled.data_ptr(), bf.data_ptr(), N, queries.data_ptr(), NQ, K, 0, out_s.data_ptr(),
                                                out_i.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), st, a, b)
        assert rc == 0, rc
        torch.cuda.synchronize()
        wall[n].append((time.perf_counter() - t0) * 1e3)
        ms = C.c_float()
        hip.hipEventSynchronize(b); hip.hipEventElapsedTime(C.byref(ms), a, b)
        times[n].append(ms.value)
        if not EXACT:
            fallbacks[n] = max(fallbacks[n], int(status[1]))
        if ref is None:
            ref = (out_s.clone(), out_i.clone())
        if not os.environ.get('AB_NOCHECK'):
            assert torch.equal(out_i, ref[1]) and torch.equal(out_s, ref[0]) and int(status[0]) == 0, n
for n in times:
    print(f"{n}: screen kernel median {np.median(times[n][1:]):.3f} ms (min {np.min(times[n][1:]):.3f}); whole call {np.median(wall[n][1:]):.3f} ms"
          + ("" if EXACT else f"; exact-fallback queries {fallbacks[n]}"), flush=True)
