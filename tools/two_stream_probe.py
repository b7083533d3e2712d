"""Same-process probe: 512 x 256 encode split into P parts over P HIP streams (library under test given
by SSKD_LIB, default the in-tree one)."""
import ctypes as C
import os
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

dev = torch.device("cuda:0")
cfg = BertConfig()
w = DeviceWeights(cfg, synthetic_state_dict(cfg), dev)
ids, mask = synthetic_ids(512, 256, cfg.vocab_size, dev)
out = torch.empty((512, 384), device=dev)
main = torch.cuda.current_stream(dev)
side = [torch.cuda.Stream(dev) for _ in range(7)]
for path in sys.argv[1:]:
    lib = C.CDLL(str(Path(path).resolve()))
    for name in ("sskd_encoder_workspace_bytes", "sskd_encoder_forward"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _native.SIGNATURES[name]
    for parts in (1, 2, 4, 8):
        n = 512 // parts
        wss = [torch.empty(int(lib.sskd_encoder_workspace_bytes(w.cstruct_cfg, n, 256)), dtype=torch.uint8, device=dev) for _ in range(parts)]

        def run():
            for p in range(1, parts):
                side[p - 1].wait_stream(main)   # fork BEFORE anything of this step is enqueued on main
            for p in range(parts):
                st = main if p == 0 else side[p - 1]
                rc = lib.sskd_encoder_forward(w.cstruct_cfg, w.struct, ids[p * n:].data_ptr(), mask[p * n:].data_ptr(), n, 256, 1,
                                              out[p * n:].data_ptr(), wss[p].data_ptr(), wss[p].numel(), int(st.cuda_stream))
                assert rc == 0
            for p in range(1, parts):
                main.wait_stream(side[p - 1])

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        print(f"{Path(path).stem}: {parts} stream(s) x {n} rows: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
