"""Screened search (bf16-MFMA screening + exact re-scoring): bit-identical to the exact scan.

Every case demands the same bits as ``oracle.search.topk_fma`` (the exact path's oracle), i.e. the
same gate as tests/test_search_gpu.py; on top, the fallback machinery is forced: duplicate-heavy
corpora (a candidate band larger than the 256 rows a query re-scores -> exact fallback inside the call, for any
number of queries), and rows / queries whose every element sits on a bf16 rounding tie with the
rounding errors of competing rows opposed (the worst case of the error band).
"""
import ctypes

import numpy as np
import pytest
import torch

from capi_helpers import stream, tile_corpus
from oracle import search as oracle
from semantic_search_kd_amd import FAISSIndexBuilder, _native

pytestmark = pytest.mark.gpu


def screened(lib, corpus, queries, k, id_offset=0):
    n, nq = corpus.shape[0], queries.shape[0]
    tiled = tile_corpus(lib, corpus)
    bf = torch.empty(int(lib.sskd_index_bf16_bytes(n)), dtype=torch.uint8, device="cuda")
    _native.check(lib.sskd_index_make_bf16(tiled.data_ptr(), n, bf.data_ptr(), stream()))
    q = torch.from_numpy(np.ascontiguousarray(queries, np.float32)).cuda()
    out_s = torch.full((nq, k), float("nan"), device="cuda")
    out_i = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
    status = torch.full((2,), -1, dtype=torch.int32, device="cuda")
    need = int(lib.sskd_index_search_screened_workspace_bytes(n, nq, k))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    _native.check(lib.sskd_index_search_screened(tiled.data_ptr(), bf.data_ptr(), n, q.data_ptr(), nq, k, id_offset,
                                                 out_s.data_ptr(), out_i.data_ptr(), status.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), stream(), None, None))
    torch.cuda.synchronize()
    return out_s.cpu().numpy(), out_i.cpu().numpy(), status.cpu().numpy()


@pytest.mark.parametrize("n,nq,k", [
    (2048, 64, 10), (2049, 65, 10), (5000, 130, 1), (20000, 300, 10), (20011, 257, 7), (100000, 1000, 10),
])
def test_screened_equals_exact_bits(gpu, native_lib, n, nq, k):
    corpus = oracle.seeded_unit_rows(n, 384, 100 + n % 97)
    queries = oracle.seeded_unit_rows(nq, 384, 200 + nq)
    # planted near neighbours make the top ranks non-trivial
    for i in range(0, nq, 7):
        queries[i] = corpus[(i * 37) % n] + 0.05 * queries[i]
        queries[i] /= np.linalg.norm(queries[i])
    s, i, st = screened(native_lib, corpus, queries, k, id_offset=1000)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, k, 1000)
    assert st[0] == 0
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)


def test_screened_unnormalised_rows_and_queries(gpu, native_lib):
    """metric = ip: the error band scales with |q| max|row|."""
    rng = np.random.default_rng(3)
    corpus = oracle.seeded_unit_rows(6000, 384, 5) * rng.uniform(0.1, 30.0, size=(6000, 1)).astype(np.float32)
    queries = oracle.seeded_unit_rows(100, 384, 6) * rng.uniform(0.01, 100.0, size=(100, 1)).astype(np.float32)
    s, i, st = screened(native_lib, corpus, queries, 10)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, 10)
    assert st[0] == 0 and np.array_equal(i, ref_i) and np.array_equal(s, ref_s)


@pytest.mark.parametrize("copies", [40, 320])
def test_screened_duplicates(gpu, native_lib, copies):
    """Consecutive copies of one row next to a quarter of the queries.  40 copies stay inside the candidate band
    (every row that reaches the pruning bound is appended: nothing is truncated - rounds 2-3 kept 6-deep sorted
    lists per lane and had to send such queries to the exact scan); 320 copies exceed the 256 candidates a query
    re-scores, so those queries are answered by the exact scan inside the call.  Either way ties resolve to the
    lower id exactly as in the exact path."""
    corpus = oracle.seeded_unit_rows(8000, 384, 11)
    corpus[3000:3000 + copies] = corpus[77]
    queries = oracle.seeded_unit_rows(128, 384, 12)
    for j in range(0, 128, 4):
        queries[j] = corpus[77] + 0.02 * queries[j]
        queries[j] /= np.linalg.norm(queries[j])
    s, i, st = screened(native_lib, corpus, queries, 10)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, 10)
    assert st[0] == 0
    assert st[1] == 0 if copies == 40 else st[1] >= 32, st   # the planted queries went through the fallback (or not)
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
    assert ref_i[0, 0] == 77 and ref_i[0, 1] == 3000  # equal scores: lower id first


def test_screened_ascending_scores_compact_their_runs(gpu, native_lib):
    """Rows whose score for the planted queries GROWS with the row id: every row is a new best, so every row
    reaches the pruning bound and a lane's run of 64 appended entries would overflow after four of the five or six
    tiles a wave owns per slice (512 tiles, 8 slices, 12 waves).  A run that
    is about to fill up first drops the entries the bound has overtaken, so these queries still come out of the
    screened path (no exact fallback) with the oracle's bits."""
    n = 16384
    u = oracle.seeded_unit_rows(1, 384, 32)[0]
    noise = oracle.seeded_unit_rows(n, 384, 33)
    noise -= np.outer(noise @ u, u)                      # orthogonal to u
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    a = np.linspace(0.05, 0.95, n, dtype=np.float32)[:, None]
    corpus = (a * u[None] + np.sqrt(1.0 - a * a) * noise).astype(np.float32)
    queries = oracle.seeded_unit_rows(4096, 384, 34)     # 32 query blocks: 8 slices, several tiles per wave
    planted = list(range(0, 4096, 2))
    for j in planted:
        queries[j] = u + 0.01 * queries[j]
        queries[j] /= np.linalg.norm(queries[j])
    s, i, st = screened(native_lib, corpus, queries, 10)
    sel = list(range(0, 4096, 64))                        # the oracle on a sample of the queries (CPU time)
    ref_s, ref_i = oracle.topk_fma(queries[sel], corpus, 10)
    assert st[0] == 0 and st[1] < len(planted) // 4, st
    assert np.array_equal(i[sel], ref_i) and np.array_equal(s[sel], ref_s)
    assert (ref_i[0] >= n - 64).all()


def test_screened_run_overflow_takes_the_exact_fallback(gpu, native_lib):
    """80 copies of a row placed where ONE lane's run must take them all (the tiles one wave of one slice owns:
    every 12th tile, 16 rows per tile and half-wave): 80 entries inside the band exceed the 64 a run holds, no
    compaction can help, the run's count records the overflow and the planted queries are answered by the exact
    scan inside the call.  (The band itself - 160 rows - would still fit the 256 candidates a query re-scores.)"""
    n, nq, waves = 65536, 1024, 12                       # 8 query blocks -> 32 slices of 64 tiles
    qpb, passes, slices = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _native.check(native_lib.sskd_index_search_screened_plan(n, nq, 10, ctypes.byref(qpb), ctypes.byref(passes),
                                                             ctypes.byref(slices)))
    tiles_per_slice = -(-(n // 32) // slices.value)
    assert tiles_per_slice >= 4 * waves + 1, (slices.value, tiles_per_slice)
    corpus = oracle.seeded_unit_rows(n, 384, 41)
    for t in range(0, 5 * waves, waves):                  # tiles 0, 12, 24, 36, 48 of slice 0: wave 0's
        corpus[32 * t : 32 * t + 32] = corpus[40000]
    queries = oracle.seeded_unit_rows(nq, 384, 42)
    planted = list(range(0, nq, 8))
    for j in planted:
        queries[j] = corpus[40000] + 0.02 * queries[j]
        queries[j] /= np.linalg.norm(queries[j])
    s, i, st = screened(native_lib, corpus, queries, 10)
    assert st[0] == 0 and st[1] >= len(planted), st
    sel = planted[:16] + [1, 2, 3]
    ref_s, ref_i = oracle.topk_fma(queries[sel], corpus, 10)
    assert np.array_equal(i[sel], ref_i) and np.array_equal(s[sel], ref_s)
    assert (ref_i[0] == np.arange(10)).all()             # ten copies, lowest ids first


def test_screened_fallback_holds_every_query(gpu, native_lib):
    """1 200 queries whose candidate band is too large to re-score (320 copies of their nearest row): the
    in-call exact fallback is sized for every query, so the C-ABI call itself returns the oracle's bits - no
    status to check, no poisoned rows (round 2 capped the fallback at 1 024 queries), on the raw entry point
    and on both product paths."""
    corpus = oracle.seeded_unit_rows(4096, 384, 21)
    corpus[1000:1320] = corpus[5]
    queries = np.repeat(corpus[5][None], 1200, axis=0) + 0.01 * oracle.seeded_unit_rows(1200, 384, 22)
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    queries = queries.astype(np.float32)
    s, i, st = screened(native_lib, corpus, queries, 10)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, 10)
    assert st[0] == 0 and st[1] > 1024
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
    index = FAISSIndexBuilder(embedding_dim=384, metric="ip", device="cuda:0")
    index.add(corpus)
    hs, hi = index.search(queries, 10)
    assert np.array_equal(hi, ref_i) and np.array_equal(hs, ref_s)
    assert index.last_search_path.endswith("+screened")
    ds, di = index.search_device(torch.from_numpy(queries).cuda(), 10)
    assert int(index.last_status[1]) > 1024
    assert np.array_equal(di.cpu().numpy(), ref_i) and np.array_equal(ds.cpu().numpy(), ref_s)


def _bf16_round(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _tie(j, e=0):
    """midpoint between the bf16 neighbours (1 + j/128) 2^e and (1 + (j+1)/128) 2^e: round-to-nearest-even
    takes it DOWN for even j and UP for odd j, with the largest relative error bf16 has (2^-8 at j = 0)"""
    return np.float32((1.0 + (j + 0.5) / 128.0) * 2.0 ** e)


def test_screened_band_covers_opposed_bf16_tie_roundings(gpu, native_lib):
    """Worst case of the error band: queries and rows whose EVERY element is a bf16 rounding tie.
    The query rounds down on coordinates 0..191 and up on 192..383.  'A' rows live on 0..191 and round down
    (both roundings lower their screen score: -0.77 %), 'B' rows live on 192..383 and round up (+0.77 %), and
    an A row's EXACT score beats the B rows' by a hair - so the screening pass ranks all ten B rows 1.5 %
    above it.  A band of half the needed width (the 0.0041 |q||c| of round 2, derived from a 2^-9 rounding
    error bf16 does not have) drops A although it is the exact top 1; asserted on the CPU below.  The
    measured band (2^-7 |q||c| here) keeps it: bit equality with the oracle."""
    n, dim = 4096, 384
    rng = np.random.default_rng(5)
    corpus = (0.01 * rng.standard_normal((n, dim))).astype(np.float32)
    a_row = np.zeros(dim, np.float32)
    a_row[:192] = _tie(2)
    a_row[7] = _tie(4)                      # lifts A's exact score just above the B rows'
    b_row = np.zeros(dim, np.float32)
    b_row[192:] = _tie(1)
    a_pos = [100 + 397 * i for i in range(3)]
    b_pos = [211 + 331 * i for i in range(10)]
    for p in a_pos:
        corpus[p] = a_row
    for p in b_pos:
        corpus[p] = b_row
    corpus[a_pos[1], 9] = _tie(4)           # A rows with distinct exact scores
    corpus[a_pos[2], 9] = _tie(4)
    corpus[a_pos[2], 11] = _tie(4)
    q = np.zeros(dim, np.float32)
    q[:192] = _tie(0)
    q[192:] = _tie(1)
    queries = np.stack([q * np.float32(2.0 ** -m) for m in range(64)]
                       + list(oracle.seeded_unit_rows(64, dim, 77))).astype(np.float32)
    k = 10
    ref_s, ref_i = oracle.topk_fma(queries, corpus, k)
    assert set(ref_i[0, :3]) == set(a_pos) and set(ref_i[0, 3:]) <= set(b_pos)
    # the case IS adversarial: with round 2's half-width band a row of the exact top k is not a candidate
    q64, c64 = _bf16_round(queries[:1]).astype(np.float64), _bf16_round(corpus).astype(np.float64)
    screen = (q64 @ c64.T)[0]
    kth = np.sort(screen)[-k]
    half_band = 2 * 0.0041 * np.linalg.norm(queries[0]) * np.linalg.norm(corpus, axis=1).max()
    assert (screen[ref_i[0]] < kth - half_band).any()
    assert (screen[ref_i[0]] >= kth - 2 * 2.0 ** -7 * 1.002 * np.linalg.norm(queries[0])
            * np.linalg.norm(corpus, axis=1).max()).all()
    s, i, st = screened(native_lib, corpus, queries, k)
    assert st[0] == 0
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
    assert st[1] < 64, st          # answered by the band, not by routing every tie query to the exact fallback


def test_product_search_device_uses_screening_and_matches_exact(gpu, native_lib):
    corpus = oracle.seeded_unit_rows(50000, 384, 31)
    queries = oracle.seeded_unit_rows(500, 384, 32)
    index = FAISSIndexBuilder(embedding_dim=384, metric="ip", device="cuda:0", id_offset=7)
    index.add(corpus[:30000])
    q = torch.from_numpy(queries).cuda()
    s1, i1 = index.search_device(q, 10)
    assert index.last_status is not None and int(index.last_status[0]) == 0
    ref = oracle.topk_fma(queries, corpus[:30000], 10, 7)
    assert np.array_equal(i1.cpu().numpy(), ref[1]) and np.array_equal(s1.cpu().numpy(), ref[0])
    index.add(corpus[30000:])   # the bf16 copy is rebuilt after an add
    s2, i2 = index.search_device(q, 10)
    ref = oracle.topk_fma(queries, corpus, 10, 7)
    assert np.array_equal(i2.cpu().numpy(), ref[1]) and np.array_equal(s2.cpu().numpy(), ref[0])
    index.screening = False
    s3, i3 = index.search_device(q, 10)
    assert index.last_status is None
    assert torch.equal(i3, i2) and torch.equal(s3, s2)
    # shapes outside the screened path (k > 10, few queries) silently take the exact scan
    index.screening = True
    s4, i4 = index.search_device(q[:10], 20)
    ref = oracle.topk_fma(queries[:10], corpus, 20, 7)
    assert np.array_equal(i4.cpu().numpy(), ref[1])


@pytest.mark.parametrize("n,nq", [(1_000_000, 10_000), (8_841_823, 1_024)])
def test_screened_equals_exact_scan_at_full_size(gpu, native_lib, n, nq):
    """BASELINE cfg 2 / cfg 3 sizes: every output row of the screened search equals the exact scan's
    (GPU vs GPU, all queries, bit for bit); the exact scan itself is held to the oracle elsewhere."""
    gen = torch.Generator(device="cuda").manual_seed(1234)
    index = FAISSIndexBuilder(embedding_dim=384, metric="ip", device="cuda:0", id_offset=11)
    index.reserve(n)
    first = None
    for lo in range(0, n, 1 << 20):
        rows = torch.randn((min(1 << 20, n - lo), 384), generator=gen, device="cuda", dtype=torch.float32)
        rows /= rows.norm(dim=1, keepdim=True)
        if first is None:
            first = rows[:4096].clone()
        index.add(rows)
    del rows
    q = torch.randn((nq, 384), generator=gen, device="cuda", dtype=torch.float32)
    q[::16] = first[: (nq + 15) // 16] + 0.02 * q[::16]     # near-duplicates of corpus rows among the queries
    q /= q.norm(dim=1, keepdim=True)
    s1, i1 = index.search_device(q, 10, normalize_queries=False)
    status = index.last_status.cpu().numpy()
    assert status[0] == 0
    index.screening = False
    s2, i2 = index.search_device(q, 10, normalize_queries=False)
    assert index.last_status is None
    assert torch.equal(i1, i2) and torch.equal(s1, s2)
    assert int(i1.min()) >= 11 and (torch.diff(s1, dim=1) <= 0).all()


def test_screened_on_anisotropic_embeddings_keeps_its_band_small(gpu, native_lib):
    """e5-like geometry: every row shares a large common component (mean pairwise cosine 0.8), 64 topical clusters
    and 1 % near-duplicate rows.  Scores of all rows then sit within +-0.03 of 0.8, and an error band proportional
    to |q| max|row| would hold hundreds of rows per query (every query -> the exact fallback).  The screening copy
    is mean-centred (q.mean is constant per query), so the band scales with the CENTRED norms: results are the
    oracle's bits and only a small share of the queries needs the fallback."""
    n, nq, dim = 100_000, 1024, 384
    rng = np.random.default_rng(8)
    common = rng.standard_normal(dim).astype(np.float32)
    common /= np.linalg.norm(common)
    centres = rng.standard_normal((64, dim)).astype(np.float32) / np.sqrt(dim)
    rows = 2.0 * common + 0.5 * centres[rng.integers(0, 64, n)] + rng.standard_normal((n, dim)).astype(np.float32) / np.sqrt(dim)
    dup = rng.integers(0, n, n // 100)
    rows[rng.integers(0, n, n // 100)] = rows[dup] + 0.01 * rng.standard_normal((n // 100, dim)).astype(np.float32) / np.sqrt(dim)
    rows = (rows / np.linalg.norm(rows, axis=1, keepdims=True)).astype(np.float32)
    q = 2.0 * common + 0.5 * centres[rng.integers(0, 64, nq)] + rng.standard_normal((nq, dim)).astype(np.float32) / np.sqrt(dim)
    q[::5] = rows[rng.integers(0, n, len(q[::5]))] + 0.1 * rng.standard_normal((len(q[::5]), dim)).astype(np.float32) / np.sqrt(dim)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    assert 0.7 < float((rows[:512] @ rows[512:1024].T).mean()) < 0.9
    s, i, st = screened(native_lib, rows, q, 10, id_offset=3)
    ref_s, ref_i = oracle.topk_fma(q, rows, 10, 3)
    assert st[0] == 0 and np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
    print("anisotropic rows: exact-fallback queries", int(st[1]), "of", nq)
    assert st[1] <= nq // 10, st


def _exact(lib, tiled, n, q, nq, k, id_offset):
    out_s = torch.empty((nq, k), device="cuda")
    out_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    ws = torch.empty(int(lib.sskd_index_search_workspace_bytes(n, nq, k)), dtype=torch.uint8, device="cuda")
    _native.check(lib.sskd_index_search(tiled.data_ptr(), n, q.data_ptr(), nq, k, id_offset, out_s.data_ptr(),
                                        out_i.data_ptr(), ws.data_ptr(), ws.numel(), stream()))
    return out_s, out_i


def test_screened_equals_exact_scan_over_random_launch_geometries(gpu, native_lib):
    """The launch plan (queries per workgroup, slices, pre-pass, XCD mapping) is a function of the shape:
    sweep shapes across its regimes - one slice, slices capped by the tile count, query blocks with a
    ragged tail, shards with and without the bound-only pre-pass - and demand the exact scan's bits (that scan is oracle-pinned in
    tests/test_search_gpu.py).  Clustered rows keep the candidate bands busy."""
    lib = native_lib
    rng = np.random.default_rng(20260)
    shapes = [(2048, 64), (2100, 256), (2050, 1000), (4000, 10000), (33000, 255), (33000, 256), (70001, 2999), (200000, 10000),
              (250000, 4100), (9000, 513), (640000, 700), (33000, 511), (33000, 512), (2048, 767), (131072 + 31, 1025)]
    for _ in range(6):
        shapes.append((int(rng.integers(2048, 300000)), int(rng.integers(64, 6000))))
    g = torch.Generator(device="cuda").manual_seed(77)
    seen = set()
    for n, nq in shapes:
        k = int(rng.integers(1, 11))
        id_offset = int(rng.integers(0, 1 << 40))
        centres = torch.randn((64, 384), generator=g, device="cuda")
        corpus = torch.nn.functional.normalize(
            centres[torch.randint(0, 64, (n,), generator=g, device="cuda")] * 0.35
            + torch.randn((n, 384), generator=g, device="cuda"), dim=1)
        queries = torch.nn.functional.normalize(
            corpus[torch.randint(0, n, (nq,), generator=g, device="cuda")] * 0.5
            + torch.randn((nq, 384), generator=g, device="cuda"), dim=1)
        qpb, passes, slices = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _native.check(lib.sskd_index_search_screened_plan(n, nq, k, ctypes.byref(qpb), ctypes.byref(passes),
                                                          ctypes.byref(slices)))
        seen.add((qpb.value, n // 32 // 8 >= 12))      # (queries per workgroup, shard large enough for the pre-pass)
        tiled = torch.zeros(int(lib.sskd_index_tiled_bytes(n)) // 4, dtype=torch.float32, device="cuda")
        _native.check(lib.sskd_index_add_rows(corpus.data_ptr(), n, 0, tiled.data_ptr(), 0, stream()))
        bf = torch.empty(int(lib.sskd_index_bf16_bytes(n)), dtype=torch.uint8, device="cuda")
        _native.check(lib.sskd_index_make_bf16(tiled.data_ptr(), n, bf.data_ptr(), stream()))
        out_s = torch.full((nq, k), float("nan"), device="cuda")
        out_i = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
        status = torch.full((2,), -1, dtype=torch.int32, device="cuda")
        ws = torch.empty(int(lib.sskd_index_search_screened_workspace_bytes(n, nq, k)), dtype=torch.uint8, device="cuda")
        _native.check(lib.sskd_index_search_screened(tiled.data_ptr(), bf.data_ptr(), n, queries.data_ptr(), nq, k, id_offset,
                                                     out_s.data_ptr(), out_i.data_ptr(), status.data_ptr(), ws.data_ptr(),
                                                     ws.numel(), stream(), None, None))
        ref_s, ref_i = _exact(lib, tiled, n, queries, nq, k, id_offset)
        torch.cuda.synchronize()
        assert int(status[0]) == 0, (n, nq, k)
        assert torch.equal(out_i, ref_i) and torch.equal(out_s, ref_s), (n, nq, k, int(status[1]))
    assert {q for q, _ in seen} == {64, 128, 160} and {d for _, d in seen} == {True, False}, seen


# --------------------------------------------------------------------------------------------------
# The sample phase and the main phase of slice 0 walk the same tiles.  A pruning pool that counted a row of those
# tiles twice would hold fewer than k DISTINCT rows behind its minimum: the "lower bound" would sit above the true
# k-th best score and prune rows of the top k silently.  These cases are built
# from the launch plan so that it would show: neighbours inside the sample rows, the rest of the top k far behind
# them in score (>> 2e) and late in the shard.
# --------------------------------------------------------------------------------------------------
def _plan(lib, n, nq, k=10):
    qpb, passes, slices = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _native.check(lib.sskd_index_search_screened_plan(n, nq, k, ctypes.byref(qpb), ctypes.byref(passes), ctypes.byref(slices)))
    return qpb.value, passes.value, slices.value


def _towards(u, cos, noise):
    """unit rows at the given cosines to the unit vector u (noise made orthogonal to u first)"""
    noise = noise - np.outer(noise @ u, u)
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    cos = np.asarray(cos, np.float32)[:, None]
    return (cos * u[None] + np.sqrt(1.0 - cos * cos) * noise).astype(np.float32)


def test_screened_sample_rows_count_once_single_slice(gpu, native_lib):
    """ONE slice (256 query blocks of 160 fill the chip): the whole sample lies inside the only slice, and the
    kernel runs its one-offer-per-lane-and-tile form.  Nine strong neighbours (cosine 0.95 ... 0.91) sit in nine
    different (tile, half-wave) groups of the sample rows - each is the best row of its group, so each is offered
    in both phases - and the tenth neighbour (0.5; then 0.47, 0.44) is among the last rows of the shard."""
    n, nq = 65536, 40960
    qpb, _, slices = _plan(native_lib, n, nq)
    assert (qpb, slices) == (160, 1), (qpb, slices)
    u = oracle.seeded_unit_rows(1, 384, 61)[0]
    corpus = oracle.seeded_unit_rows(n, 384, 62)
    groups = [(3, 0), (9, 1), (14, 0), (20, 1), (27, 0), (33, 1), (41, 0), (50, 1), (58, 0)]   # (tile < 64, half)
    near = [32 * t + 4 * h + 1 for t, h in groups]
    far = [n - 37, n - 1500, n - 4001]
    corpus[near] = _towards(u, [0.95 - 0.005 * i for i in range(9)], oracle.seeded_unit_rows(9, 384, 63))
    corpus[far] = _towards(u, [0.5, 0.47, 0.44], oracle.seeded_unit_rows(3, 384, 64))
    queries = oracle.seeded_unit_rows(nq, 384, 65)
    planted = list(range(5, nq, 160))
    queries[planted] = _towards(u, [0.999] * len(planted), queries[planted])
    s, i, st = screened(native_lib, corpus, queries, 10)
    sel = planted[:24] + planted[-8:] + [0, 1, 2, nq - 1]
    ref_s, ref_i = oracle.topk_fma(queries[sel], corpus, 10)
    assert list(ref_i[0]) == near + far[:1]
    assert st[0] == 0
    assert np.array_equal(i[sel], ref_i) and np.array_equal(s[sel], ref_s), (i[sel[0]], ref_i[0], int(st[1]))


@pytest.mark.parametrize("members", [5, 7, 9])
def test_screened_near_duplicate_cluster_from_row_zero(gpu, native_lib, members):
    """5-9 near-duplicates of one passage in the shard's first rows (consecutive rows from row 0, and a second
    cluster with one member per tile), the other neighbours of the top 10 much weaker and at the far end; a
    several-slice geometry whose slice 0 still contains the sample."""
    n, nq = 131072, 2048
    u = oracle.seeded_unit_rows(2, 384, 71)
    corpus = oracle.seeded_unit_rows(n, 384, 72)
    c0 = list(range(members))                               # consecutive rows from row 0
    c1 = [32 * (2 * t + 1) + 5 for t in range(members)]     # one per odd tile, upper half-wave
    corpus[c0] = _towards(u[0], [0.97 - 0.004 * i for i in range(members)], oracle.seeded_unit_rows(members, 384, 73))
    corpus[c1] = _towards(u[1], [0.96 - 0.004 * i for i in range(members)], oracle.seeded_unit_rows(members, 384, 74))
    tail0 = [n - 100 - 977 * i for i in range(8)]
    tail1 = [n // 2 + 31 + 1013 * i for i in range(8)]
    corpus[tail0] = _towards(u[0], [0.55 - 0.03 * i for i in range(8)], oracle.seeded_unit_rows(8, 384, 75))
    corpus[tail1] = _towards(u[1], [0.52 - 0.03 * i for i in range(8)], oracle.seeded_unit_rows(8, 384, 76))
    queries = oracle.seeded_unit_rows(nq, 384, 77)
    p0, p1 = list(range(0, nq, 16)), list(range(7, nq, 16))
    queries[p0] = _towards(u[0], [0.999] * len(p0), queries[p0])
    queries[p1] = _towards(u[1], [0.999] * len(p1), queries[p1])
    s, i, st = screened(native_lib, corpus, queries, 10)
    sel = p0[:12] + p1[:12] + [1, 2, 3]
    ref_s, ref_i = oracle.topk_fma(queries[sel], corpus, 10)
    assert list(ref_i[0]) == c0 + tail0[: 10 - members] and list(ref_i[12]) == c1 + tail1[: 10 - members]
    assert st[0] == 0
    assert np.array_equal(i[sel], ref_i) and np.array_equal(s[sel], ref_s)


def test_screened_topic_sorted_corpus_at_the_bench_geometry(gpu, native_lib):
    """Locality-ordered corpus at the cfg-2 shape (1 M rows x 10 000 queries: 63 blocks of 160 queries x 4 slices,
    the every-appended-row-is-offered form): rows sorted by topic, a topic = documents of 8 adjacent chunks.  A
    query about a document finds its 8 chunks (0.9) and then the best chunks of OTHER documents of the topic (0.6),
    thousands of rows later.  Queries on every document of the shard's first 2 048 rows (the sample), of the rows
    around the slice boundaries and of the last rows; the oracle's bits on those, the exact scan's on all."""
    n, nq, per_doc, topics = 1_000_000, 10_000, 8, 64
    qpb, _, slices = _plan(native_lib, n, nq)
    assert (qpb, slices) == (160, 4), (qpb, slices)
    rng = np.random.Generator(np.random.PCG64(81))
    n_docs = n // per_doc
    docs_per_topic = -(-n_docs // topics)
    centres = oracle.seeded_unit_rows(topics, 384, 82)
    topic_of_doc = np.arange(n_docs) // docs_per_topic
    doc = 0.7 * centres[topic_of_doc] + 0.7 * oracle.seeded_unit_rows(n_docs, 384, 83)
    doc /= np.linalg.norm(doc, axis=1, keepdims=True)
    corpus = np.repeat(doc, per_doc, axis=0)
    corpus *= np.float32(0.9)
    corpus += np.float32(0.44) * oracle.seeded_unit_rows(n, 384, 84)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    corpus = corpus.astype(np.float32)
    tiles_per_slice = -(-(n // 32) // slices)
    asked = list(range(0, 2048 // per_doc))                                            # the sample rows
    for sl in range(1, slices):
        asked += list(range(sl * tiles_per_slice * 32 // per_doc - 4, sl * tiles_per_slice * 32 // per_doc + 4))
    asked += list(range(n_docs - 8, n_docs))
    queries = oracle.seeded_unit_rows(nq, 384, 85)
    planted = [(37 * j) % nq for j in range(len(asked))]
    assert len(set(planted)) == len(planted)
    queries[planted] = doc[asked] + 0.1 * queries[planted]
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    queries = queries.astype(np.float32)

    tiled = tile_corpus(native_lib, corpus)
    bf = torch.empty(int(native_lib.sskd_index_bf16_bytes(n)), dtype=torch.uint8, device="cuda")
    _native.check(native_lib.sskd_index_make_bf16(tiled.data_ptr(), n, bf.data_ptr(), stream()))
    q = torch.from_numpy(queries).cuda()
    out_s = torch.full((nq, 10), float("nan"), device="cuda")
    out_i = torch.full((nq, 10), -7, dtype=torch.int64, device="cuda")
    status = torch.full((2,), -1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(native_lib.sskd_index_search_screened_workspace_bytes(n, nq, 10)), dtype=torch.uint8, device="cuda")
    _native.check(native_lib.sskd_index_search_screened(tiled.data_ptr(), bf.data_ptr(), n, q.data_ptr(), nq, 10, 0,
                                                        out_s.data_ptr(), out_i.data_ptr(), status.data_ptr(), ws.data_ptr(),
                                                        ws.numel(), stream(), None, None))
    ex_s, ex_i = _exact(native_lib, tiled, n, q, nq, 10, 0)
    torch.cuda.synchronize()
    assert int(status[0]) == 0
    ref_s, ref_i = oracle.topk_fma(queries[planted], corpus, 10)
    own = np.array([[per_doc * d + c for c in range(per_doc)] for d in asked])
    assert all(set(own[r]) <= set(ref_i[r]) for r in range(len(asked)))               # 8 own chunks + 2 of the topic
    got_i, got_s = out_i.cpu().numpy(), out_s.cpu().numpy()
    wrong = [asked[r] for r in range(len(asked)) if not (np.array_equal(got_i[planted[r]], ref_i[r])
                                                          and np.array_equal(got_s[planted[r]], ref_s[r]))]
    assert not wrong, ("documents whose query lost a top-10 row", wrong[:20], len(wrong))
    assert torch.equal(out_i, ex_i) and torch.equal(out_s, ex_s)


def test_index_is_one_row_major_copy_and_the_sidecar_holds_no_second_one(gpu, native_lib):
    """Round 4: the index is the plain row-major fp32 matrix (zero-padded to 32 rows) and serves scan, exact
    re-scoring and save(); the screening sidecar is the bf16 tiles + a 4-KiB norm block - index + sidecar = 1.5x the
    corpus (rounds 2-3: 2.5x, a second fp32 copy lived in the sidecar)."""
    lib = native_lib
    for n in (32, 2049, 1_000_000, 8_841_823):
        padded = int(lib.sskd_index_padded_rows(n))
        assert int(lib.sskd_index_tiled_bytes(n)) == padded * 1536
        assert int(lib.sskd_index_bf16_bytes(n)) == padded * 768 + 4096
    corpus = oracle.seeded_unit_rows(2049, 384, 9)
    buf = tile_corpus(lib, corpus)
    got = buf.cpu().numpy().reshape(-1, 384)
    assert got.shape[0] == 2080 and np.array_equal(got[:2049], corpus) and not got[2049:].any()
