"""Does the two-branch trick of the encoder forward (csrc/encoder.hip sskd_encoder_forward) carry over to the teacher
cross-encoder?  128 pairs x 256 tokens as one launch against two halves of 64 pairs on two HIP streams.
``python tools/two_stream_teacher_probe.py`` prints ms per 128 pairs."""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from semantic_search_kd_amd import _native  # noqa: E402
from semantic_search_kd_amd.teacher import TeacherConfig, TeacherModel  # noqa: E402

dev = torch.device("cuda:0")
cfg = TeacherConfig()
t = TeacherModel.from_random_device(cfg, "cuda:0", seed=0)
lib = _native.load()
P, S = 128, 256
g = torch.Generator(device=dev).manual_seed(5)
ids = torch.randint(4, cfg.vocab_size, (P, S), generator=g, device=dev, dtype=torch.int32)
ids[:, 0] = 0
ids[:, -1] = 2
mask = torch.ones_like(ids)
out = torch.empty(P, dtype=torch.float32, device=dev)


def ws_for(b):
    return torch.empty(int(lib.sskd_teacher_workspace_bytes(t._cfg, b, S)), dtype=torch.uint8, device=dev)


ws_full, ws_a, ws_b = ws_for(P), ws_for(P // 2), ws_for(P // 2)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def score(lo, n, ws, stream):
    _native.check(lib.sskd_teacher_score(t._cfg, t._w, *t._head, ids[lo:lo + n].data_ptr(), mask[lo:lo + n].data_ptr(), n, S,
                                         out[lo:lo + n].data_ptr(), ws.data_ptr(), ws.numel(), int(stream.cuda_stream)))


def one():
    score(0, P, ws_full, s1)


def two():
    score(0, P // 2, ws_a, s1)
    score(P // 2, P // 2, ws_b, s2)


def timeit(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one()
torch.cuda.synchronize()
ref = out.clone()
for name, f in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
    ms = timeit(f)
    print(f"{name}: {ms:.2f} ms per {P} pairs, max |diff| vs one stream {float((out - ref).abs().max()):.2e}", flush=True)
