"""Experiment (round 4): does running the two halves of a 512 x 256 batch on two HIP streams, half a layer apart,
overlap the fused MLP's un-hidden memory phases (its prologue / epilogue bursts) with the other half's compute?
``python tools/two_stream_probe.py`` prints ms per 512-document forward: one stream, two streams in
lockstep, two streams with the second one delayed by ~0.25 ms."""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from semantic_search_kd_amd import _native  # noqa: E402
from semantic_search_kd_amd.bench_support import synthetic_ids  # noqa: E402
from semantic_search_kd_amd.weights import BertConfig, DeviceWeights, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = BertConfig()
w = DeviceWeights(cfg, synthetic_state_dict(cfg), dev)
lib = _native.load()
B, S = 512, 256
ids, mask = synthetic_ids(B, S, cfg.vocab_size, dev)
out = torch.empty((B, 384), dtype=torch.float32, device=dev)


def ws_for(b):
    return torch.empty(int(lib.sskd_encoder_workspace_bytes(w.cstruct_cfg, b, S)), dtype=torch.uint8, device=dev)


ws_full, ws_a, ws_b = ws_for(B), ws_for(B // 2), ws_for(B // 2)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def fwd(lo, n, ws, stream):
    rc = lib.sskd_encoder_forward(w.cstruct_cfg, w.struct, ids[lo:lo + n].data_ptr(), mask[lo:lo + n].data_ptr(), n, S, 1,
                                  out[lo:lo + n].data_ptr(), ws.data_ptr(), ws.numel(), int(stream.cuda_stream))
    assert rc == 0, rc


def one_stream():
    fwd(0, B, ws_full, s1)


def two_streams(delay_cycles):
    def go():
        fwd(0, B // 2, ws_a, s1)
        with torch.cuda.stream(s2):
            if delay_cycles:
                torch.cuda._sleep(delay_cycles)
        fwd(B // 2, B // 2, ws_b, s2)
    return go


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one_stream()
torch.cuda.synchronize()
ref = out.clone()
for name, f in (("one stream", one_stream), ("two streams, lockstep", two_streams(0)),
                ("two streams, second delayed ~0.12 ms", two_streams(250_000)),
                ("two streams, second delayed ~0.25 ms", two_streams(500_000)),
                ("one stream", one_stream)):
    ms = timeit(f)
    torch.cuda.synchronize()
    print(f"{name}: {ms:.3f} ms per 512 documents ({B / ms:.1f} k docs/s), max |diff| vs one stream {float((out - ref).abs().max()):.2e}", flush=True)
