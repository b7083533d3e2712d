"""ctypes binding of ``libsskd_amd.so`` (the C-ABI declared in ``include/sskd_amd.h``).

There is deliberately no fallback: if the HIP library is missing or a call fails
the caller gets an exception, never a CPU path.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

_LIB_PATH = Path(__file__).resolve().parent / "libsskd_amd.so"

SSKD_OK = 0
SSKD_DIM = 384
SSKD_TILE_ROWS = 32
SSKD_K_PASS = 32
SSKD_K_MAX = 1024

_ERR_NAMES = {1: "INVALID", 2: "WORKSPACE", 3: "HIP", 4: "UNSUPPORTED"}


class NativeError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"sskd_amd[{_ERR_NAMES.get(code, code)}]: {message}")
        self.code = code


class EncoderConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32),
        ("hidden", C.c_int32),
        ("layers", C.c_int32),
        ("heads", C.c_int32),
        ("intermediate", C.c_int32),
        ("max_positions", C.c_int32),
        ("type_vocab", C.c_int32),
        ("layer_norm_eps", C.c_float),
    ]


class EncoderLayerWeights(C.Structure):
    _fields_ = [
        (n, C.c_void_p)
        for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")
    ]


class EncoderWeights(C.Structure):
    _fields_ = [
        ("word_emb", C.c_void_p),
        ("pos_emb", C.c_void_p),
        ("type_emb", C.c_void_p),
        ("emb_ln_g", C.c_void_p),
        ("emb_ln_b", C.c_void_p),
        ("layers", C.POINTER(EncoderLayerWeights)),
    ]


class GenericConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
        ("intermediate", C.c_int32), ("max_positions", C.c_int32), ("type_vocab", C.c_int32),
        ("layer_norm_eps", C.c_float), ("pos_offset", C.c_int32),
    ]


GENERIC_LAYER_W = ("wqkv", "wqkv_t", "wo", "wo_t", "w1", "w1_t", "w2", "w2_t",
                   "bqkv", "bo", "ln1_g", "ln1_b", "b1", "b2", "ln2_g", "ln2_b")
GENERIC_LAYER_G = ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")


class GenericLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GENERIC_LAYER_W]


class GenericWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b")] + [
        ("layers", C.POINTER(GenericLayerWeights))
    ]


class GenericLayerGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GENERIC_LAYER_G]


class GenericGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b")] + [
        ("layers", C.POINTER(GenericLayerGrads))
    ]


class SearchTuning(C.Structure):
    """``sskd_search_tuning``: explicit launch tuning, passed to the workspace query AND the search."""

    _fields_ = [("queries_per_block", C.c_int32), ("target_workgroups", C.c_int32), ("pruning_pools", C.c_int32)]


# name -> (restype, argtypes); must list every symbol of include/sskd_amd.h
_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_ip = C.POINTER(C.c_int)
_f = C.c_float
SIGNATURES = {
    "sskd_abi_version": (_i, []),
    "sskd_last_error": (C.c_char_p, []),
    "sskd_device_count": (_i, []),
    "sskd_index_padded_rows": (_i64, [_i64]),
    "sskd_index_tiled_bytes": (_sz, [_i64]),
    "sskd_index_add_rows": (_i, [_vp, _i64, _i, _vp, _i64, _vp]),
    "sskd_index_get_rows": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "sskd_l2_normalize_rows": (_i, [_vp, _i64, _i, _vp]),
    "sskd_index_search_workspace_bytes": (_sz, [_i64, _i, _i]),
    "sskd_index_search": (_i, [_vp, _i64, _vp, _i, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sskd_index_search_profiled": (_i, [_vp, _i64, _vp, _i, _i, _i64, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "sskd_index_search_plan": (_i, [_i64, _i, _i, _ip, _ip, _ip, _ip, _ip]),
    "sskd_index_search_workspace_bytes_ex": (_sz, [_i64, _i, _i, C.POINTER(SearchTuning)]),
    "sskd_index_search_ex": (
        _i, [_vp, _i64, _vp, _i, _i, _i64, _vp, _vp, _vp, _sz, _vp, C.POINTER(SearchTuning), _vp, _vp]
    ),
    "sskd_index_search_plan_ex": (_i, [_i64, _i, _i, C.POINTER(SearchTuning), _ip, _ip, _ip, _ip, _ip]),
    "sskd_topk_record_bytes": (_sz, [_i, _i]),
    "sskd_topk_merge_packed": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sskd_index_bf16_bytes": (_sz, [_i64]),
    "sskd_index_make_bf16": (_i, [_vp, _i64, _vp, _vp]),
    "sskd_index_search_screened_workspace_bytes": (_sz, [_i64, _i, _i]),
    "sskd_index_search_screened_plan": (_i, [_i64, _i, _i, _ip, _ip, _ip]),
    "sskd_index_search_screened": (_i, [_vp, _vp, _i64, _vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "sskd_index_search_onepass_workspace_bytes": (_sz, [_i64, _i, _i]),
    "sskd_index_search_onepass": (_i, [_vp, _i64, _vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sskd_topk_merge": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sskd_kd_loss": (_i, [_vp, _vp, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "sskd_similarity": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "sskd_pool_normalize": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "sskd_encoder_workspace_bytes": (_sz, [C.POINTER(EncoderConfig), _i, _i]),
    "sskd_encoder_forward": (
        _i,
        [C.POINTER(EncoderConfig), C.POINTER(EncoderWeights), _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp],
    ),
    "sskd_pack_plan": (_i, [_vp, _i, _i, _vp, _ip]),
    "sskd_pack_tokens": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sskd_encoder_forward_packed": (
        _i,
        [C.POINTER(EncoderConfig), C.POINTER(EncoderWeights), _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp],
    ),
    "sskd_generic_workspace_bytes": (_sz, [C.POINTER(GenericConfig), _i, _i, _i]),
    "sskd_generic_forward": (
        _i, [C.POINTER(GenericConfig), C.POINTER(GenericWeights), _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]
    ),
    "sskd_generic_backward": (
        _i, [C.POINTER(GenericConfig), C.POINTER(GenericWeights), C.POINTER(GenericGrads), _vp, _vp, _i, _i, _i, _vp,
             _vp, _sz, _vp]
    ),
    "sskd_teacher_workspace_bytes": (_sz, [C.POINTER(GenericConfig), _i, _i]),
    "sskd_teacher_score": (
        _i, [C.POINTER(GenericConfig), C.POINTER(GenericWeights), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]
    ),
    "sskd_gemm_nt_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sskd_gemm_backend": (_i, [_i]),
    "sskd_gemm_tn_bf16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "sskd_tokenizer_create": (_i, [C.c_char_p, _i64, C.POINTER(C.c_void_p)]),
    "sskd_tokenizer_destroy": (None, [_vp]),
    "sskd_tokenizer_encode": (_i, [_vp, C.c_char_p, _vp, _i, _i, _i, _vp, _i64, _vp, _vp, C.POINTER(_i64)]),
    "sskd_encoder_hidden": (
        _i,
        [C.POINTER(EncoderConfig), C.POINTER(EncoderWeights), _vp, _vp, _i, _i, _vp, _vp, _sz, _vp],
    ),
}

_lib: Optional[C.CDLL] = None


def lib_path() -> Path:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the C-ABI library, binding argtypes/restypes. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise ImportError(
            f"{_LIB_PATH} is missing: the MI355X HIP extension is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != SSKD_OK:
        msg = load().sskd_last_error()
        raise NativeError(code, msg.decode("utf-8", "replace") if msg else "unknown error")


def require_gpu() -> None:
    """Fail loudly when no HIP device is usable (product classes call this)."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "semantic-search-kd_amd is an MI355X-only backend: no HIP device is visible "
            "(torch.cuda.is_available() is False) and there is no CPU path."
        )
    load()


def current_stream_ptr(device=None) -> int:
    import torch

    return int(torch.cuda.current_stream(device).cuda_stream)
