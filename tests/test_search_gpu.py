"""GPU parity of the exact top-k scan: HIP (through the C-ABI) vs the oracle, bit for bit.

Oracle order: ``oracle.search.topk_fma`` sums each dot product in the kernel's fma order, so
scores AND ids must be identical.  ``topk_blas`` (numpy ``q @ c.T`` + argsort, the reference
idiom src/kd/eval.py:86) is the second witness: scores within 1e-3 (north star tolerance; in
practice < 1e-6) and identical ids outside near-ties.
"""
import ctypes

import numpy as np
import pytest
import torch

from capi_helpers import capi_search, stream, tile_corpus
from oracle import search as oracle
from semantic_search_kd_amd import FAISSIndexBuilder, _native

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-3  # BASELINE.json north_star: "cosine scores within 1e-3 fp32"


def _check_exact(lib, corpus, queries, k, id_offset=0):
    tiled = tile_corpus(lib, corpus)
    s, i = capi_search(lib, tiled, corpus.shape[0], queries, k, id_offset)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, k, id_offset)
    assert np.array_equal(i, ref_i)
    assert np.array_equal(s, ref_s)
    return s, i


def test_index_layout_roundtrip_is_bit_exact(gpu, native_lib):
    for n in (1, 31, 32, 33, 1000, 4099):
        x = oracle.seeded_unit_rows(n, 384, 7 + n)
        tiled = tile_corpus(native_lib, x)
        back = torch.empty((n, 384), dtype=torch.float32, device="cuda")
        _native.check(native_lib.sskd_index_get_rows(tiled.data_ptr(), 0, n, back.data_ptr(), stream()))
        assert np.array_equal(back.cpu().numpy(), x)
        # documented layout (round 4): the plain row-major matrix, zero-padded to a multiple of 32 rows
        padded = ((n + 31) // 32) * 32
        t = tiled.cpu().numpy()[: padded * 384].reshape(padded, 384)
        assert np.array_equal(t[:n], x)
        assert not t[n:].any()
        # a sub-range read back from the middle
        if n > 40:
            part = torch.empty((7, 384), dtype=torch.float32, device="cuda")
            _native.check(native_lib.sskd_index_get_rows(tiled.data_ptr(), 33, 7, part.data_ptr(), stream()))
            assert np.array_equal(part.cpu().numpy(), x[33:40])
    # normalising add: x / ||x||, zero rows untouched (faiss.normalize_L2)
    x = oracle.seeded_unit_rows(70, 384, 3) * np.linspace(0.1, 9.0, 70, dtype=np.float32)[:, None]
    x[5] = 0
    t = tile_corpus(native_lib, x, normalize=True).cpu().numpy()[: 96 * 384].reshape(96, 384)
    want = oracle.l2_normalize_rows(x)
    assert np.allclose(t[:70], want, rtol=0, atol=2e-7) and not t[5].any() and not t[70:].any()


def test_conftest_recipe_10_docs(gpu, native_lib):
    """The reference's only executable index fixture: seed-42 randn(10, 384) unit rows in an
    IndexFlatIP (tests/conftest.py:65-73,184-185); k in {1, 3, 10, 20 -> padded}."""
    np.random.seed(42)
    emb = np.random.randn(10, 384).astype(np.float32)
    emb = emb / np.linalg.norm(emb, axis=1, keepdims=True)
    q = oracle.seeded_unit_rows(5, 384, 99)
    q[0] = emb[3]  # a query equal to a document must retrieve it first with score ~1
    for k in (1, 3, 10, 20):
        s, i = _check_exact(native_lib, emb, q, k)
        assert i[0, 0] == 3 and abs(s[0, 0] - 1.0) < 1e-6
        if k > 10:
            assert (i[:, 10:] == -1).all()
            assert (s[:, 10:] == np.finfo(np.float32).min).all()


@pytest.mark.parametrize("tag", ["small", "1k"])
def test_hip_search_matches_reference_code_fixture(gpu, native_lib, tag):
    """HIP scan vs ids / scores produced by the reference's own evaluate_model / KDEvaluator
    (tests/golden/search_ref_*.npz, see make_golden.make_search_ref): identical ids, scores
    within the north-star tolerance (observed ~1e-7)."""
    from conftest import GOLDEN

    gold = np.load(GOLDEN / f"search_ref_{tag}.npz")
    if tag == "small":
        q, c, ks = gold["queries"], gold["corpus"], (1, 3, 10, 20)
    else:
        q, c, ks = oracle.seeded_unit_rows(100, 384, 4321), oracle.seeded_unit_rows(1000, 384, 1234), (1, 5, 10)
    near = set(gold["near_tie_queries"].tolist())
    firm = np.array([i for i in range(q.shape[0]) if i not in near])
    tiled = tile_corpus(native_lib, c)
    for k in ks:
        s, i = capi_search(native_lib, tiled, c.shape[0], q, k)
        ref_i, ref_s = gold[f"ref_ids_k{k}"], gold[f"ref_scores_k{k}"]
        kk = ref_i.shape[1]
        assert np.array_equal(i[firm, :kk], ref_i[firm])
        assert np.abs(s[:, :kk] - ref_s).max() <= SCORE_TOL
        assert np.abs(s[:, :kk] - ref_s).max() <= 1e-6
        assert (i[:, kk:] == -1).all()


@pytest.mark.parametrize(
    "n,nq,k",
    [
        (1000, 100, 10),   # BASELINE cfg 1
        (1, 1, 1),
        (31, 3, 5),
        (32, 32, 10),
        (33, 33, 10),
        (1003, 65, 10),    # ragged corpus, ragged query blocks (QB=2 path)
        (4099, 97, 16),
        (5000, 7, 32),
        (777, 2, 10),
        (20000, 130, 10),
    ],
)
def test_search_bit_exact_vs_oracle(gpu, native_lib, n, nq, k):
    corpus = oracle.seeded_unit_rows(n, 384, 1234 + n)
    queries = oracle.seeded_unit_rows(nq, 384, 4321 + nq)
    s, i = _check_exact(native_lib, corpus, queries, k)
    # second witness: BLAS-order scores of the reference idiom
    bs, bi = oracle.topk_blas(queries, corpus, k)
    valid = i >= 0
    assert np.abs(s[valid] - bs[valid]).max() <= SCORE_TOL
    ties = set(oracle.near_tie_queries(oracle.topk_blas(queries, corpus, min(k + 1, n))[0]).tolist())
    for qi in range(nq):
        if qi not in ties:
            assert np.array_equal(i[qi], bi[qi])
    # rows come back sorted by (score desc, id asc)
    assert (np.diff(s, axis=1) <= 0).all()


@pytest.mark.parametrize("k", [33, 50, 100, 200])
def test_large_k_chained_passes(gpu, native_lib, k):
    """k above one scan pass (schemas.py:12-16 allows k <= 100, rerank_top_k <= 200)."""
    corpus = oracle.seeded_unit_rows(3000, 384, 11)
    queries = oracle.seeded_unit_rows(5, 384, 12)
    _check_exact(native_lib, corpus, queries, k)
    # fewer rows than k: tail is padding
    _check_exact(native_lib, corpus[:70], queries, k)


@pytest.mark.parametrize("nq,k", [(1, 10), (1, 50), (1, 100), (1, 200), (3, 33), (40, 64), (64, 100), (65, 100)])
def test_online_shapes_few_queries_many_slices(gpu, native_lib, nq, k):
    """The /search shape (schemas.py:12-16: one query, k <= 100, rerank_top_k <= 200): a few queries
    are spread over hundreds of slices, so the group-reduce steps and (for k > 32 with <= 64
    queries) the chained K = 10 passes carry the result."""
    n = 150_000
    corpus = oracle.seeded_unit_rows(n, 384, 77)
    queries = oracle.seeded_unit_rows(nq, 384, 78)
    # plant near-duplicates so that several of the best rows share a per-lane list
    corpus[1000:1012] = oracle.l2_normalize_rows(queries[0][None, :] + 0.05 * corpus[1000:1012])
    _check_exact(native_lib, corpus, queries, k)
    plan = [ctypes.c_int() for _ in range(5)]
    _native.check(native_lib.sskd_index_search_plan(n, nq, k, *[ctypes.byref(x) for x in plan]))
    assert plan[2].value >= 64  # many slices


def _onepass(lib, tiled, n, queries, k, id_offset=0):
    nq = queries.shape[0]
    q = torch.from_numpy(np.ascontiguousarray(queries, np.float32)).cuda()
    out_s = torch.full((nq, k), float("nan"), dtype=torch.float32, device="cuda")
    out_i = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
    flag = torch.full((1,), 9, dtype=torch.int32, device="cuda")
    ws = torch.empty(max(int(lib.sskd_index_search_onepass_workspace_bytes(n, nq, k)), 1), dtype=torch.uint8, device="cuda")
    _native.check(
        lib.sskd_index_search_onepass(
            tiled.data_ptr(), n, q.data_ptr(), nq, k, id_offset, out_s.data_ptr(), out_i.data_ptr(),
            flag.data_ptr(), ws.data_ptr(), ws.numel(), stream(),
        )
    )
    torch.cuda.synchronize()
    return out_s.cpu().numpy(), out_i.cpu().numpy(), int(flag.item())


@pytest.mark.parametrize("nq,k", [(1, 33), (1, 50), (1, 100), (1, 200), (1, 256), (5, 64), (64, 100)])
def test_onepass_large_k_is_proven_exact_on_ordinary_data(gpu, native_lib, nq, k):
    """One corpus pass + proof (sskd_amd.h): on ordinary data the best rows are spread over the
    per-lane lists, the proof succeeds, and the result is bit-identical to the oracle."""
    n = 150_000
    corpus = oracle.seeded_unit_rows(n, 384, 91)
    queries = oracle.seeded_unit_rows(nq, 384, 92)
    tiled = tile_corpus(native_lib, corpus)
    s, i, inexact = _onepass(native_lib, tiled, n, queries, k, id_offset=1000)
    assert inexact == 0
    ref_s, ref_i = oracle.topk_fma(queries, corpus, k, 1000)
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)


def test_onepass_refuses_to_certify_when_a_list_overflows(gpu, native_lib):
    """Sixteen near-copies of the query inside ONE per-lane list (rows 0-3, 8-11, 16-19, 24-27 of
    one tile = one wave, one half): the list keeps ten, the proof must fail, and the host path falls
    back to the chained search - the answer is exact either way."""
    n = 60_000
    corpus = oracle.seeded_unit_rows(n, 384, 93)
    queries = oracle.seeded_unit_rows(2, 384, 94)
    tile0 = 32 * 777
    rows = [tile0 + (r & 3) + 8 * (r >> 2) for r in range(16)]
    corpus[rows] = oracle.l2_normalize_rows(queries[0][None, :] + 0.02 * corpus[rows])
    tiled = tile_corpus(native_lib, corpus)
    _, _, inexact = _onepass(native_lib, tiled, n, queries, 50)
    assert inexact == 1
    # 300 rows, k = 256: every list saw 16 rows of a tile and kept ten - not certifiable either
    small = corpus[:300]
    _, _, inexact = _onepass(native_lib, tile_corpus(native_lib, small), 300, queries, 256)
    assert inexact == 1
    # 8 rows: no list is full, nothing was dropped, always proven; k beyond the row count pads
    tiny = corpus[:8]
    s, i, inexact = _onepass(native_lib, tile_corpus(native_lib, tiny), 8, queries, 40)
    assert inexact == 0
    ref_s, ref_i = oracle.topk_fma(queries, tiny, 40)
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
    # the class falls back by itself
    ib = FAISSIndexBuilder(384, "Flat", "ip")
    ib.add(torch.from_numpy(corpus).cuda())
    s, i = ib.search(queries, 50)
    assert ib.last_search_path == "onepass-unproven+chained"
    ref_s, ref_i = oracle.topk_fma(queries, corpus, 50)
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
    s, i = ib.search(queries[1:], 50)
    assert ib.last_search_path == "onepass"
    assert np.array_equal(i, ref_i[1:]) and np.array_equal(s, ref_s[1:])


def test_host_search_randomised_small_shapes_with_ties(gpu):
    """FAISSIndexBuilder.search() (one pass + proof, chained fallback) on 40 random shapes: row counts
    around tile boundaries, 1-64 queries, k up to 256 and beyond the row count, corpora made largely
    of duplicated rows (exact score ties) - always bit-identical to the oracle, whichever path ran."""
    rng = np.random.default_rng(2024)
    paths = set()
    for case in range(40):
        n = int(rng.choice([1, 7, 31, 32, 33, 100, 257, 1000, 4096, 5000]))
        nq = int(rng.choice([1, 2, 5, 32, 33, 64]))
        k = int(rng.choice([1, 5, 10, 11, 32, 33, 64, 100, 200, 256]))
        base = oracle.seeded_unit_rows(max(n // 3, 1), 384, 1000 + case)
        corpus = base[rng.integers(0, base.shape[0], size=n)]          # ~3 copies of every row
        queries = oracle.seeded_unit_rows(nq, 384, 2000 + case)
        queries[0] = corpus[rng.integers(0, n)]                         # an exact hit with its copies
        ib = FAISSIndexBuilder(384, "Flat", "ip")
        ib.add(torch.from_numpy(np.ascontiguousarray(corpus)).cuda())
        s, i = ib.search(queries, k)
        ref_s, ref_i = oracle.topk_fma(queries, corpus, k)
        assert np.array_equal(i, ref_i), (case, n, nq, k, ib.last_search_path)
        assert np.array_equal(s, ref_s), (case, n, nq, k, ib.last_search_path)
        paths.add(ib.last_search_path)
    assert "onepass" in paths


def test_ties_resolve_to_lower_id(gpu, native_lib):
    base = oracle.seeded_unit_rows(40, 384, 5)
    corpus = np.concatenate([base, base, base[:7]])  # every row appears 2-3 times -> exact score ties
    queries = base[:9].copy()
    s, i = _check_exact(native_lib, corpus, queries, 10)
    for qi in range(9):
        # the three copies of the query itself come first, lowest id first
        assert i[qi, 0] == qi and i[qi, 1] == qi + 40
    # constant corpus: all scores equal -> ids 0..k-1
    const = np.tile(base[:1], (100, 1))
    s, i = _check_exact(native_lib, const, queries[:2], 10)
    assert np.array_equal(i[0], np.arange(10))


@pytest.mark.parametrize("nq", [1, 33, 65])
def test_all_negative_scores_with_padded_query_lanes(gpu, native_lib, nq):
    """Padding queries score 0 on every row; that bound must never leak into a real query."""
    d = oracle.seeded_unit_rows(1, 384, 21)
    queries = oracle.l2_normalize_rows(d + 0.3 * oracle.seeded_unit_rows(nq, 384, 22))
    corpus = oracle.l2_normalize_rows(-d + 0.3 * oracle.seeded_unit_rows(20000, 384, 23))
    s, _ = _check_exact(native_lib, corpus, queries, 10)
    assert (s < 0).all()
    _check_exact(native_lib, corpus[:3000], queries, 40)


def test_many_exact_ties_across_workgroups(gpu, native_lib):
    """Each score occurs 20 times, spread over every slice: the shared pools must keep ties."""
    base = oracle.seeded_unit_rows(1000, 384, 31)
    corpus = np.tile(base, (20, 1))
    queries = oracle.seeded_unit_rows(70, 384, 32)
    s, i = _check_exact(native_lib, corpus, queries, 10)
    assert (np.diff(i, axis=1) == 1000).all()  # the ten lowest-id copies of the best row
    _check_exact(native_lib, corpus, queries[:5], 50)


def test_adversarial_ascending_scores(gpu, native_lib):
    """Every row beats all earlier ones (worst case for the running threshold)."""
    q = oracle.seeded_unit_rows(1, 384, 3)
    noise = oracle.seeded_unit_rows(2000, 384, 4)
    alpha = np.linspace(0.0, 0.9, 2000, dtype=np.float32)[:, None]
    corpus = oracle.l2_normalize_rows(alpha * q + (1 - alpha) * noise)
    s, i = _check_exact(native_lib, corpus, np.repeat(q, 3, axis=0), 10)
    assert i[0, 0] == 1999
    # and descending
    _check_exact(native_lib, corpus[::-1].copy(), q, 10)


def test_id_offset_and_empty_inputs(gpu, native_lib):
    corpus = oracle.seeded_unit_rows(300, 384, 21)
    queries = oracle.seeded_unit_rows(4, 384, 22)
    s, i = _check_exact(native_lib, corpus, queries, 10, id_offset=1_105_228)
    assert i.min() >= 1_105_228
    # empty corpus: all padding
    tiled = torch.zeros(1, device="cuda")
    s, i = capi_search(native_lib, tiled, 0, queries, 5)
    assert (i == -1).all() and (s == np.finfo(np.float32).min).all()
    # zero queries: nothing written, no error
    s, i = capi_search(native_lib, tile_corpus(native_lib, corpus), 300, queries[:0], 5)
    assert s.shape == (0, 5)


def test_special_values(gpu, native_lib):
    corpus = oracle.seeded_unit_rows(200, 384, 31)
    corpus[17, 5] = np.nan       # NaN score is never selected
    corpus[18] = 0.0             # zero row scores exactly 0
    queries = oracle.seeded_unit_rows(3, 384, 32)
    s, i = _check_exact(native_lib, corpus, queries, 10)
    assert 17 not in i
    assert not np.isnan(s).any()


def test_workspace_too_small_is_an_error(gpu, native_lib):
    corpus = oracle.seeded_unit_rows(64, 384, 41)
    tiled = tile_corpus(native_lib, corpus)
    q = torch.zeros((1, 384), device="cuda")
    out_s = torch.empty((1, 5), device="cuda")
    out_i = torch.empty((1, 5), dtype=torch.int64, device="cuda")
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")
    rc = native_lib.sskd_index_search(
        tiled.data_ptr(), 64, q.data_ptr(), 1, 5, 0, out_s.data_ptr(), out_i.data_ptr(), ws.data_ptr(), 16, stream()
    )
    assert rc == 2 and b"workspace" in native_lib.sskd_last_error()


def test_topk_merge_matches_oracle_and_unsharded(gpu, native_lib):
    """K9: per-shard partial top-k (global ids) merged == search over the whole corpus."""
    corpus = oracle.seeded_unit_rows(2500, 384, 51)
    queries = oracle.seeded_unit_rows(37, 384, 52)
    k = 10
    bounds = [0, 800, 801, 1700, 2500]  # ragged shards, one with a single row
    parts = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        tiled = tile_corpus(native_lib, corpus[lo:hi])
        parts.append(capi_search(native_lib, tiled, hi - lo, queries, k, id_offset=lo))
    ps = np.stack([p[0] for p in parts])
    pi = np.stack([p[1] for p in parts])
    d_s, d_i = torch.from_numpy(ps).cuda(), torch.from_numpy(pi).cuda()
    out_s = torch.empty((37, k), device="cuda")
    out_i = torch.empty((37, k), dtype=torch.int64, device="cuda")
    _native.check(
        native_lib.sskd_topk_merge(d_s.data_ptr(), d_i.data_ptr(), len(parts), 37, k, k, out_s.data_ptr(), out_i.data_ptr(), stream())
    )
    ms, mi = oracle.topk_merge(ps, pi, k)
    assert np.array_equal(out_i.cpu().numpy(), mi) and np.array_equal(out_s.cpu().numpy(), ms)
    whole_s, whole_i = oracle.topk_fma(queries, corpus, k)
    assert np.array_equal(mi, whole_i) and np.array_equal(ms, whole_s)


def test_l2_normalize_and_similarity(gpu, native_lib):
    g = np.random.default_rng(0)
    x = (g.standard_normal((70, 384)) * 3).astype(np.float32)
    x[5] = 0.0
    d = torch.from_numpy(x).cuda()
    _native.check(native_lib.sskd_l2_normalize_rows(d.data_ptr(), 70, 384, stream()))
    got = d.cpu().numpy()
    np.testing.assert_allclose(got, oracle.l2_normalize_rows(x), rtol=0, atol=2e-7)
    assert not got[5].any()
    # compute_similarity (eval.py:75): q @ d.T, same fma order as the scan -> bit exact
    q = oracle.seeded_unit_rows(45, 384, 1)
    c = oracle.seeded_unit_rows(131, 384, 2)
    out = torch.empty((45, 131), device="cuda")
    dq, dc = torch.from_numpy(q).cuda(), torch.from_numpy(c).cuda()
    _native.check(native_lib.sskd_similarity(dq.data_ptr(), 45, dc.data_ptr(), 131, 384, out.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.scores_fma(q, c))
    np.testing.assert_allclose(out.cpu().numpy(), oracle.scores_blas(q, c), atol=1e-6)


def test_builder_class_search_save_load(gpu, tmp_path):
    """FAISSIndexBuilder-shaped surface: cosine metric normalises rows and queries."""
    g = np.random.default_rng(3)
    raw = (g.standard_normal((530, 384)) * 2).astype(np.float32)  # NOT unit norm
    queries = (g.standard_normal((6, 384))).astype(np.float32)
    b = FAISSIndexBuilder(embedding_dim=384, index_type="HNSW", metric="cosine")
    b.add(raw[:100])
    b.add(raw[100:117])   # leaves a partial tail tile
    b.add(raw[117:])      # re-packs it
    assert b.index.ntotal == 530
    b.doc_ids = [f"chunk_{i}" for i in range(530)]
    D, I = b.search(queries, k=10)
    assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (6, 10)
    ref_s, ref_i = oracle.topk_blas(oracle.l2_normalize_rows(queries), oracle.l2_normalize_rows(raw), 10)
    assert np.array_equal(I, ref_i)
    np.testing.assert_allclose(D, ref_s, atol=1e-5)
    # single 1-d query like the serving route would pass after encode
    D1, I1 = b.search(queries[0], k=3)
    assert I1.shape == (1, 3) and np.array_equal(I1[0], ref_i[0, :3])
    b.save(tmp_path / "idx")
    assert (tmp_path / "idx" / "index.faiss").exists() and (tmp_path / "idx" / "doc_ids.json").exists()
    b2 = FAISSIndexBuilder(embedding_dim=384)
    b2.load(tmp_path / "idx")
    assert b2.doc_ids[:2] == ["chunk_0", "chunk_1"] and b2.index.ntotal == 530
    D2, I2 = b2.search(queries, k=10)
    assert np.array_equal(I2, I) and np.array_equal(D2, D)


def test_full_size_properties_1m(gpu, native_lib):
    """BASELINE cfg 2 size (1M x 384, 10k queries, k=10) through size-independent properties:
    planted neighbours are found at rank 1, rows are sorted, ids valid and unique, and a
    256-query subsample is bit-exact against the oracle."""
    n, nq, k = 1_000_000, 10_000, 10
    gen = torch.Generator(device="cuda").manual_seed(1234)
    rows = torch.randn((n, 384), generator=gen, device="cuda", dtype=torch.float32)
    rows /= rows.norm(dim=1, keepdim=True)
    qgen = torch.Generator(device="cuda").manual_seed(4321)
    queries = torch.randn((nq, 384), generator=qgen, device="cuda", dtype=torch.float32)
    # plant: every 10th query is a noisy copy of corpus row 97*i
    planted = torch.arange(0, nq, 10, device="cuda")
    targets = (planted * 97) % n
    queries[planted] = rows[targets] + 0.3 * queries[planted] / 384 ** 0.5
    queries /= queries.norm(dim=1, keepdim=True)
    b = FAISSIndexBuilder(embedding_dim=384, metric="ip")
    b.add(rows)
    s, i = b.search_device(queries, k)
    torch.cuda.synchronize()
    s, i = s.cpu().numpy(), i.cpu().numpy()
    assert (i >= 0).all() and (i < n).all()
    assert (np.diff(s, axis=1) <= 0).all()
    assert all(len(set(r)) == k for r in i[:500])
    assert np.array_equal(i[planted.cpu().numpy(), 0], targets.cpu().numpy())
    sub = np.arange(0, nq, nq // 256)[:256]
    ref_s, ref_i = oracle.topk_fma(queries[sub].cpu().numpy(), rows.cpu().numpy(), k)
    assert np.array_equal(i[sub], ref_i) and np.array_equal(s[sub], ref_s)
    # idempotence: a second search returns the same bits
    s2, i2 = b.search_device(queries, k)
    assert np.array_equal(s2.cpu().numpy(), s) and np.array_equal(i2.cpu().numpy(), i)


def test_cfg3_full_corpus_8_8m_rows(gpu, native_lib):
    """BASELINE cfg 3 size on one GPU: 8 841 823 rows (13.6 GB in HBM), checked bit-exact against the
    oracle on a handful of queries, plus one shard of the 8-way split with its id offset."""
    n = 8_841_823
    gen = torch.Generator(device="cuda").manual_seed(1234)
    b = FAISSIndexBuilder(embedding_dim=384, metric="ip")
    b.reserve(n)
    host_chunks = []
    for lo in range(0, n, 1 << 20):
        hi = min(lo + (1 << 20), n)
        rows = torch.randn((hi - lo, 384), generator=gen, device="cuda", dtype=torch.float32)
        rows /= rows.norm(dim=1, keepdim=True)
        b.add(rows)
        host_chunks.append(rows.cpu().numpy())
    assert b.index.ntotal == n
    corpus = np.concatenate(host_chunks)
    del host_chunks
    q = oracle.seeded_unit_rows(6, 384, 77)
    q[0] = corpus[n - 1]          # the very last row (ragged final tile) must be retrievable
    q[1] = corpus[4_420_911]
    D, I = b.search(q, k=10)
    ref_s, ref_i = oracle.topk_fma(q, corpus, 10)
    assert np.array_equal(I, ref_i) and np.array_equal(D, ref_s)
    assert I[0, 0] == n - 1 and I[1, 0] == 4_420_911
    # shard 7 of 8 (1 105 227 rows) with global ids
    from semantic_search_kd_amd.dist import shard_bounds

    lo, hi = shard_bounds(n, 8, 7)
    shard = FAISSIndexBuilder(embedding_dim=384, metric="ip", id_offset=lo)
    shard.add(torch.from_numpy(corpus[lo:hi]).cuda())
    Ds, Is = shard.search(q, k=10)
    ref_s, ref_i = oracle.topk_fma(q, corpus[lo:hi], 10, id_offset=lo)
    assert np.array_equal(Is, ref_i) and np.array_equal(Ds, ref_s) and Is.min() >= lo


def test_packed_merge_equals_plain_merge(gpu, native_lib):
    """sskd_topk_merge_packed over the records ONE all-gather moves == sskd_topk_merge; odd nq * k
    exercises the record padding."""
    from semantic_search_kd_amd.dist import hip_merge_packed, record_bytes, record_views

    corpus = oracle.seeded_unit_rows(1500, 384, 61)
    for nq, k in ((37, 10), (3, 5), (1, 1)):
        queries = oracle.seeded_unit_rows(nq, 384, 62 + nq)
        bounds = [0, 500, 501, 1500]
        g = len(bounds) - 1
        rec = record_bytes(nq, k)
        assert rec == native_lib.sskd_topk_record_bytes(nq, k) and rec % 16 == 0
        records = torch.zeros(g * rec, dtype=torch.uint8, device="cuda")
        for r, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
            s, i = capi_search(native_lib, tile_corpus(native_lib, corpus[lo:hi]), hi - lo, queries, k, id_offset=lo)
            vs, vi = record_views(records[r * rec : (r + 1) * rec], nq, k)
            vs.copy_(torch.from_numpy(s))
            vi.copy_(torch.from_numpy(i))
        out_s, out_i = hip_merge_packed(records, g, nq, k, k)
        whole_s, whole_i = oracle.topk_fma(queries, corpus, k)
        assert np.array_equal(out_i.cpu().numpy(), whole_i) and np.array_equal(out_s.cpu().numpy(), whole_s)


def test_search_tuning_struct_changes_plan_not_results(gpu, native_lib):
    """The launch tuning is an explicit argument handed to the workspace query AND the search (no
    environment knobs); every setting returns the same bits."""
    corpus = oracle.seeded_unit_rows(20000, 384, 71)
    queries = oracle.seeded_unit_rows(130, 384, 72)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, 10)
    index = FAISSIndexBuilder(embedding_dim=384, metric="ip", device="cuda:0")
    index.add(corpus)
    q = torch.from_numpy(queries).cuda()
    seen = set()
    for tn in (None, _native.SearchTuning(32, 0, 0), _native.SearchTuning(64, 16, 1), _native.SearchTuning(0, 2048, -1)):
        index.search_tuning = tn
        s, i = index.search_device(q, 10, normalize_queries=False)
        assert np.array_equal(i.cpu().numpy(), ref_i) and np.array_equal(s.cpu().numpy(), ref_s)
        plan = [ctypes.c_int() for _ in range(5)]
        _native.check(native_lib.sskd_index_search_plan_ex(20000, 130, 10, tn, *plan))
        seen.add((plan[0].value, plan[2].value))
    assert len(seen) >= 3  # the tuning really changed queries-per-block / slices
