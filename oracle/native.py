"""ctypes loader for the plain-C oracle (``oracle/csrc/oracle.c``). Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = _DIR / "_build" / "liboracle.so"
_lib = None


def build(force: bool = False) -> Path:
    src = _DIR / "csrc" / "oracle.c"
    if force or not _LIB.exists() or _LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "-B" if force else "-s"], check=True)
    return _LIB


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(str(_LIB))
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        lib.oracle_scores_fma.argtypes = [f32p, C.c_int, f32p, C.c_int64, C.c_int, f32p]
        lib.oracle_search_fma.argtypes = [f32p, C.c_int, f32p, C.c_int64, C.c_int, C.c_int, C.c_int64, f32p, i64p]
        lib.oracle_topk_of_scores.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_int64, f32p, i64p]
        lib.oracle_topk_merge.argtypes = [f32p, i64p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, i64p]
        lib.oracle_l2_normalize_rows.argtypes = [f32p, C.c_int64, C.c_int]
        lib.oracle_pool_normalize.argtypes = [f32p, i32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        for fn in (
            lib.oracle_scores_fma, lib.oracle_search_fma, lib.oracle_topk_of_scores,
            lib.oracle_topk_merge, lib.oracle_l2_normalize_rows, lib.oracle_pool_normalize,
        ):
            fn.restype = None
        _lib = lib
    return _lib
