"""Search oracle (CPU) — TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Two restatements of the reference's exact search, which must agree with each other:

``topk_blas``  the reference idiom verbatim: ``scores = q @ corpus.T`` (scripts/simple_eval.py:25,
               src/kd/eval.py:75) then ``np.argsort(scores)[::-1][:k]`` (src/kd/eval.py:86), made
               deterministic by the build's tie rule (equal scores: lower id first).
``topk_fma``   the same search with each score summed in the gfx950 kernel's fma order
               (plain C, ``oracle/csrc/oracle.c``) — comparable bit for bit with the HIP path.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import native

NEG_PAD = np.float32(-3.4028234663852886e38)  # faiss CMin<float>::neutral() = lowest()


def l2_normalize_rows(x: np.ndarray) -> np.ndarray:
    """``faiss.normalize_L2`` (configs/index.yaml:30): rows of zero norm are left untouched."""
    out = np.ascontiguousarray(x, dtype=np.float32).copy()
    native.load().oracle_l2_normalize_rows(out, out.shape[0], out.shape[1])
    return out


def scores_blas(q: np.ndarray, c: np.ndarray) -> np.ndarray:
    """``np.matmul(query_embs, corpus_embs.T)`` — scripts/simple_eval.py:25."""
    return np.matmul(np.asarray(q, np.float32), np.asarray(c, np.float32).T)


def scores_fma(q: np.ndarray, c: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q, np.float32)
    c = np.ascontiguousarray(c, np.float32)
    out = np.empty((q.shape[0], c.shape[0]), np.float32)
    native.load().oracle_scores_fma(q, q.shape[0], c, c.shape[0], c.shape[1], out)
    return out


def topk_of_scores(scores: np.ndarray, k: int, id_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Pure-numpy ``argsort(scores)[::-1][:k]`` with the tie rule, padded like faiss (-1 ids)."""
    scores = np.asarray(scores, np.float32)
    nq, n = scores.shape
    out_s = np.full((nq, k), NEG_PAD, np.float32)
    out_i = np.full((nq, k), -1, np.int64)
    ids = np.arange(n, dtype=np.int64)
    for i in range(nq):
        s = scores[i]
        valid = ~(np.isnan(s) | np.isneginf(s))
        # lexsort: last key is primary -> descending score, then ascending id
        order = np.lexsort((ids[valid], -s[valid].astype(np.float64)))[:k]
        sel = ids[valid][order]
        out_s[i, : len(sel)] = s[sel]
        out_i[i, : len(sel)] = sel + id_offset
    return out_s, out_i


def topk_blas(q: np.ndarray, c: np.ndarray, k: int, id_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    if c.shape[0] == 0:
        nq = np.asarray(q).shape[0]
        return np.full((nq, k), NEG_PAD, np.float32), np.full((nq, k), -1, np.int64)
    return topk_of_scores(scores_blas(q, c), k, id_offset)


def topk_fma(q: np.ndarray, c: np.ndarray, k: int, id_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    q = np.ascontiguousarray(q, np.float32)
    c = np.ascontiguousarray(c, np.float32).reshape(-1, q.shape[1])
    out_s = np.empty((q.shape[0], k), np.float32)
    out_i = np.empty((q.shape[0], k), np.int64)
    native.load().oracle_search_fma(q, q.shape[0], c, c.shape[0], q.shape[1], k, id_offset, out_s, out_i)
    return out_s, out_i


def topk_merge(scores: np.ndarray, ids: np.ndarray, k_out: int) -> Tuple[np.ndarray, np.ndarray]:
    """Merge per-shard lists ``[n_lists, nq, k_in]`` into ``[nq, k_out]`` (after the all-gather)."""
    scores = np.ascontiguousarray(scores, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    n_lists, nq, k_in = scores.shape
    out_s = np.empty((nq, k_out), np.float32)
    out_i = np.empty((nq, k_out), np.int64)
    native.load().oracle_topk_merge(scores, ids, n_lists, nq, k_in, k_out, out_s, out_i)
    return out_s, out_i


def pool_normalize(hidden: np.ndarray, mask: np.ndarray, normalize: bool = True) -> np.ndarray:
    """Masked mean-pool + L2-normalise (sentence-transformers Pooling(mean) + Normalize)."""
    hidden = np.ascontiguousarray(hidden, np.float32)
    mask = np.ascontiguousarray(mask, np.int32)
    B, S, H = hidden.shape
    out = np.empty((B, H), np.float32)
    native.load().oracle_pool_normalize(hidden, mask, B, S, H, int(normalize), out)
    return out


def near_tie_queries(scores_sorted: np.ndarray, gap: float = 2e-6) -> np.ndarray:
    """Queries whose consecutive oracle scores in the top (k+1) are closer than ``gap``.

    fp32 summation order differs between BLAS and the GPU; ranks inside such a gap are
    legitimately interchangeable (SURVEY.md §7 "exact-id parity").
    """
    d = scores_sorted[:, :-1] - scores_sorted[:, 1:]
    return np.where((d < gap).any(axis=1))[0]


def seeded_unit_rows(n: int, dim: int, seed: int) -> np.ndarray:
    """Synthetic corpus / query generator of BASELINE.md §4: N(0,1) rows, L2-normalised."""
    g = np.random.Generator(np.random.PCG64(seed))
    x = g.standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)
