"""TFLOP/s of the generic NT GEMM (``sskd_gemm_nt_bf16``) on the shapes of the KD step and the teacher:
``python tools/gemm_probe.py [lib.so ...]`` (same-box A/B when several libraries are given)."""
import ctypes as C
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from semantic_search_kd_amd import _native  # noqa: E402

SHAPES = [  # (M, N, K, c_is_f32, accumulate, label)
    (65536, 1152, 384, 0, 0, "student QKV"),
    (65536, 384, 384, 0, 0, "student out-proj"),
    (65536, 1536, 384, 0, 0, "student FFN1"),
    (65536, 384, 1536, 0, 0, "student FFN2"),
    (384, 384, 65536, 1, 1, "student dWo (split-K)"),
    (1152, 384, 65536, 1, 1, "student dWqkv (split-K)"),
    (1536, 384, 65536, 1, 1, "student dW1 (split-K)"),
    (384, 1536, 65536, 1, 1, "student dW2 (split-K)"),
    (32768, 3072, 1024, 0, 0, "teacher QKV"),
    (32768, 1024, 1024, 0, 0, "teacher out-proj"),
    (32768, 4096, 1024, 0, 0, "teacher FFN1"),
    (32768, 1024, 4096, 0, 0, "teacher FFN2"),
    (8192, 8192, 8192, 0, 0, "8192^3"),
]
dev = torch.device("cuda:0")
paths = sys.argv[1:] or [str(_native._LIB_PATH)]
libs = []
for pth in paths:
    lib = C.CDLL(str(Path(pth).resolve()))
    fn = lib.sskd_gemm_nt_bf16
    fn.restype, fn.argtypes = _native.SIGNATURES["sskd_gemm_nt_bf16"]
    if hasattr(lib, "sskd_gemm_backend"):   # both routes of one library: its own kernels only, then automatic (hipBLASLt
        be = lib.sskd_gemm_backend          # for the plain large-K products)
        be.restype, be.argtypes = _native.SIGNATURES["sskd_gemm_backend"]
        libs.append((Path(pth).stem + "[own]", lambda *a, fn=fn, be=be: (be(1), fn(*a))[1]))
        libs.append((Path(pth).stem + "[auto]", lambda *a, fn=fn, be=be: (be(0), fn(*a))[1]))
    else:
        libs.append((Path(pth).stem, fn))
st = int(torch.cuda.current_stream(dev).cuda_stream)
for M, N, K, f32, acc, label in SHAPES:
    a = (torch.rand((M, K), device=dev) - 0.5).to(torch.bfloat16)
    b = (torch.rand((N, K), device=dev) - 0.5).to(torch.bfloat16)
    c = torch.zeros((M, N), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    line = f"{label:24s} M={M:6d} N={N:5d} K={K:6d}"
    for name, fn in libs:
        for _ in range(3):
            assert fn(a.data_ptr(), b.data_ptr(), c.data_ptr(), None if acc else bias.data_ptr(), M, N, K, f32, acc, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            fn(a.data_ptr(), b.data_ptr(), c.data_ptr(), None if acc else bias.data_ptr(), M, N, K, f32, acc, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        line += f" | {name}: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF/s"
        if not acc and M * N <= 1 << 28:   # sampled rows against torch in fp32 (the probe doubles as a smoke check of a variant)
            rows = torch.randint(0, M, (512,), device=dev)
            ref = a[rows].float() @ b.float().T
            err = ((c[rows].float() - ref).abs().max() / ref.abs().max()).item()
            line += f" err {err:.1e}" + ("" if err < 1.5e-2 else " MISMATCH")
    # what the vendor library reaches on the same product (torch.nn.functional.linear -> hipBLASLt / rocBLAS): a yardstick, not
    # a path of the product
    if not f32:
        import torch.nn.functional as F

        bb = bias.to(torch.bfloat16)
        for _ in range(3):
            F.linear(a, b, bb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            F.linear(a, b, bb)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f" | torch linear: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF/s"
    if f32 and acc:   # weight-gradient shape: the vendor library on C[M, N] = A[K, M]^T B[K, N] (token-major operands, as dW has them)
        at = (torch.rand((K, M), device=dev) - 0.5).to(torch.bfloat16)
        bt = (torch.rand((K, N), device=dev) - 0.5).to(torch.bfloat16)
        for _ in range(3):
            torch.mm(at.t(), bt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.mm(at.t(), bt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f" | torch mm(A^T, B) bf16 out: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF/s"
    print(line, flush=True)
