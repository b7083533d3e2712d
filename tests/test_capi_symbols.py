"""The C-ABI library loads and exports every symbol that include/sskd_amd.h declares (no GPU)."""
import re

from conftest import REPO


def _declared_symbols():
    text = (REPO / "include" / "sskd_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sskd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert "sskd_index_search" in syms and "sskd_encoder_forward" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol(native_lib):
    from semantic_search_kd_amd import _native

    declared = _declared_symbols()
    for name in declared:
        assert hasattr(native_lib, name), f"libsskd_amd.so does not export {name}"
    # and the ctypes table binds exactly the declared set
    assert sorted(_native.SIGNATURES) == declared


def test_abi_version_and_pure_host_queries(native_lib):
    assert native_lib.sskd_abi_version() == 1
    assert native_lib.sskd_index_padded_rows(0) == 0
    assert native_lib.sskd_index_padded_rows(1) == 32
    assert native_lib.sskd_index_padded_rows(1_000_000) == 1_000_000
    assert native_lib.sskd_index_padded_rows(8_841_823) == 8_841_824
    assert native_lib.sskd_index_tiled_bytes(1_000_000) == 1_000_000 * 384 * 4
    assert native_lib.sskd_index_search_workspace_bytes(1_000_000, 10_000, 10) > 0


def test_generic_workspace_covers_the_two_branch_split(native_lib):
    """sskd_generic_forward / _backward / sskd_teacher_score cut batches of >= 2 x 16 384 tokens into two halves on two streams,
    each half carving its own workspace out of the caller's buffer (csrc/train.hip generic_parts: one rule on (cfg, B, S,
    training), so forward and backward agree).  The size query must cover both halves - and must not charge small batches."""
    import ctypes as C

    from semantic_search_kd_amd import _native

    cfg = _native.GenericConfig(vocab_size=30522, hidden=384, layers=12, heads=12, intermediate=1536, max_positions=512,
                                type_vocab=2, layer_norm_eps=1e-12, pos_offset=0)
    ws = lambda B, S, tr: int(native_lib.sskd_generic_workspace_bytes(C.byref(cfg), B, S, tr))
    for tr in (0, 1):
        whole, half = ws(256, 256, tr), ws(128, 256, tr)         # 2 x 32 768 tokens: split
        assert half > 0 and whole >= 2 * half
        assert ws(128, 256, tr) >= 2 * ws(64, 256, tr)           # 2 x 16 384 tokens: the smallest batch that splits
        small, smaller = ws(64, 256, tr), ws(32, 256, tr)        # 2 x 8 192 tokens: one part, sized as before
        assert smaller < small < 2 * smaller + 4096 * 64
        assert ws(3, 64, tr) > 0 and ws(0, 256, tr) == 0
    t = lambda B, S: int(native_lib.sskd_teacher_workspace_bytes(C.byref(cfg), B, S))
    assert t(128, 256) == ws(128, 256, 0)


def test_search_plan_reports_geometry(native_lib):
    import ctypes as C

    qpb, passes, slices, waves, scans = (C.c_int() for _ in range(5))
    rc = native_lib.sskd_index_search_plan(1_000_000, 10_000, 10, qpb, passes, slices, waves, scans)
    assert rc == 0
    assert qpb.value in (32, 64) and waves.value == 8 and scans.value == 1
    assert passes.value == -(-10_000 // qpb.value)
    assert slices.value % 8 == 0
    # k > 32 is served by chained passes: 32 results per pass for many queries, 10 per pass (the
    # light kernel) when there are at most 64 queries
    rc = native_lib.sskd_index_search_plan(1_000_000, 10_000, 100, qpb, passes, slices, waves, scans)
    assert rc == 0 and scans.value == 4 and qpb.value == 32
    rc = native_lib.sskd_index_search_plan(1000, 1, 100, qpb, passes, slices, waves, scans)
    assert rc == 0 and scans.value == 10 and qpb.value == 32


def test_tuning_struct_is_explicit_and_sizes_the_workspace(native_lib):
    """No hidden process-global knobs: the plan depends only on the arguments (and the struct)."""
    import ctypes as C
    import os

    from semantic_search_kd_amd import _native

    base = native_lib.sskd_index_search_workspace_bytes(1_000_000, 10_000, 10)
    os.environ["SSKD_SCAN_QB"] = "1"  # round-1 knob: must be ignored now
    try:
        assert native_lib.sskd_index_search_workspace_bytes(1_000_000, 10_000, 10) == base
    finally:
        del os.environ["SSKD_SCAN_QB"]
    assert native_lib.sskd_index_search_workspace_bytes_ex(1_000_000, 10_000, 10, None) == base
    tn = _native.SearchTuning(32, 0, 0)
    qpb, passes = C.c_int(), C.c_int()
    assert native_lib.sskd_index_search_plan_ex(1_000_000, 10_000, 10, tn, qpb, passes, None, None, None) == 0
    assert qpb.value == 32 and passes.value == 313
    wide = _native.SearchTuning(0, 4096, 0)  # more slices -> more partial lists -> larger workspace
    assert native_lib.sskd_index_search_workspace_bytes_ex(1_000_000, 10_000, 10, wide) > base
    assert native_lib.sskd_topk_record_bytes(10_000, 10) == 1_200_000
    assert native_lib.sskd_topk_record_bytes(3, 5) == 192  # 180 B padded to 16


def test_invalid_arguments_are_reported_not_thrown(native_lib):
    rc = native_lib.sskd_index_search(None, 10, None, 1, 0, 0, None, None, None, 0, None)
    assert rc == 1
    assert b"k=0" in native_lib.sskd_last_error()
    rc = native_lib.sskd_index_add_rows(None, 5, 0, None, 7, None)
    assert rc == 1


def test_product_refuses_cpu_device():
    import pytest
    from semantic_search_kd_amd import FAISSIndexBuilder

    with pytest.raises(RuntimeError, match="MI355X only"):
        FAISSIndexBuilder(embedding_dim=384, device="cpu")
