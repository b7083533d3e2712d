"""Time of the TN weight-gradient GEMM (``sskd_gemm_tn_bf16``) on the student's dW shapes, next to the
NT kernel on pre-transposed operands (``sskd_gemm_nt_bf16``, split-K): ``python tools/gemm_tn_probe.py [lib.so ...]``
(same-box A/B when several libraries are given; every TN result is checked against torch in fp32)."""
import ctypes as C
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from semantic_search_kd_amd import _native  # noqa: E402

dev = torch.device("cuda:0")
st = int(torch.cuda.current_stream(dev).cuda_stream)
libs = []
for pth in sys.argv[1:] or [str(_native._LIB_PATH)]:
    lib = C.CDLL(str(Path(pth).resolve()))
    for name in ("sskd_gemm_tn_bf16", "sskd_gemm_nt_bf16"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = _native.SIGNATURES[name]
    libs.append((Path(pth).stem, lib))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for T in (65536, 8192, 1024):
    for M, N, label in ((384, 384, "dWo"), (1152, 384, "dWqkv"), (1536, 384, "dW1"), (384, 1536, "dW2")):
        a = (torch.rand((T, M), device=dev) - 0.5).to(torch.bfloat16)
        b = (torch.rand((T, N), device=dev) - 0.5).to(torch.bfloat16)
        ref = a.float().T @ b.float()
        fl = 2.0 * M * N * T
        line = f"T={T:6d} {label:6s} M={M:5d} N={N:5d}"
        for name, lib in libs:
            c = torch.zeros((M, N), device=dev)
            assert lib.sskd_gemm_tn_bf16(a.data_ptr(), b.data_ptr(), c.data_ptr(), T, M, N, st) == 0
            err = ((c - ref).abs().max() / ref.abs().max()).item()
            t_tn = timed(lambda: lib.sskd_gemm_tn_bf16(a.data_ptr(), b.data_ptr(), c.data_ptr(), T, M, N, st))
            line += f" | {name}: {t_tn:7.1f} us {fl / t_tn / 1e6:6.1f} TF/s err {err:.0e}" + ("" if err < 1e-3 else " MISMATCH")
        if T == 65536:
            at, bt = a.T.contiguous(), b.T.contiguous()
            c = torch.zeros((M, N), device=dev)
            t_nt = timed(lambda: libs[0][1].sskd_gemm_nt_bf16(at.data_ptr(), bt.data_ptr(), c.data_ptr(), None, M, N, T, 1, 1, st))
            line += f" | NT on transposed operands {t_nt:7.1f} us"
        print(line, flush=True)
