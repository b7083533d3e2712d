"""Row-sharded search over world_size-2 process groups.

CPU (``gloo``): the N > 1 plumbing of ``ShardedSearcher`` — shard bounds, global id offsets, the
packed per-rank record, ONE all-gather and the merge — with the HIP calls replaced by the oracle
(tests may use the oracle; the product default is ``search_device`` + ``sskd_topk_merge_packed``).

GPU (``-m gpu``): the PRODUCT path end to end in two processes — ``FAISSIndexBuilder.search_device``
writing into the packed record, the all-gather and the HIP packed merge.  On a box with >= 2 GPUs
the collective is RCCL (backend ``nccl``); on the one-GPU test box both ranks share cuda:0 (RCCL
refuses two ranks on one device), so the records cross through a host-staged ``gloo`` all-gather —
everything else is the code bench.py runs.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search as oracle
from semantic_search_kd_amd.dist import ShardedSearcher, shard_bounds


def test_shard_bounds_cover_the_corpus_exactly():
    for n, g in ((1_000_000, 8), (8_841_823, 8), (10, 4), (3, 8), (0, 2), (1000, 1)):
        spans = [shard_bounds(n, g, r) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # 8 841 823 rows over 8 GPUs -> 1 105 228 per rank, last shard 1 105 227
    assert shard_bounds(8_841_823, 8, 0) == (0, 1_105_228)
    assert shard_bounds(8_841_823, 8, 7) == (7 * 1_105_228, 8_841_823)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, nq, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        corpus = oracle.seeded_unit_rows(n, 384, 1234)
        queries = torch.from_numpy(oracle.seeded_unit_rows(nq, 384, 4321))
        lo, hi = shard_bounds(n, world, rank)

        def local_search(q, kk):
            s, i = oracle.topk_fma(q.numpy(), corpus[lo:hi], kk, id_offset=lo)
            return torch.from_numpy(s), torch.from_numpy(i)

        def merge(all_s, all_i, kk):
            s, i = oracle.topk_merge(all_s.numpy(), all_i.numpy(), kk)
            return torch.from_numpy(s), torch.from_numpy(i)

        searcher = ShardedSearcher(local_search, merge=merge)
        s, i = searcher.search(queries, k)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), s=s.numpy(), i=i.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,nq,k,world", [(1000, 17, 10, 2), (5, 3, 10, 2), (1003, 9, 10, 4), (3, 2, 5, 4)])
def test_sharded_search_equals_unsharded(tmp_path, n, nq, k, world):
    """world 4 with a ragged last shard, and more ranks than rows (empty shards) as well"""
    mp.spawn(_worker, args=(world, _free_port(), n, nq, k, str(tmp_path)), nprocs=world, join=True)
    corpus = oracle.seeded_unit_rows(n, 384, 1234)
    queries = oracle.seeded_unit_rows(nq, 384, 4321)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, k)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["i"], ref_i) and np.array_equal(got["s"], ref_s)


def _score_worker(rank, world, port, n, out_dir):
    """teacher-style data parallelism (cfg 5): contiguous item ranges, ONE all-gather as the final concat"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semantic_search_kd_amd.dist import sharded_scores

        calls = []

        def score(lo, hi):
            calls.append((lo, hi))
            return torch.arange(lo, hi, dtype=torch.float32) * 0.5 - 3.0

        out = sharded_scores(score, n)
        assert calls == ([shard_bounds(n, world, rank)] if shard_bounds(n, world, rank)[1] > shard_bounds(n, world, rank)[0] else [])
        np.save(os.path.join(out_dir, f"scores{rank}.npy"), out.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(1000, 2), (7, 2), (1, 2), (10, 4), (3, 4)])
def test_sharded_scores_concat_equals_unsharded(tmp_path, n, world):
    mp.spawn(_score_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    want = np.arange(n, dtype=np.float32) * 0.5 - 3.0
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"scores{r}.npy"), want)


def _sharded_case(n, nq, dups):
    """Seeded rows / queries; ``dups``: 320 copies of one row inside EACH shard and every query next to it -
    more than 1 024 queries per rank whose candidate band holds more rows than the screened search re-scores
    (256), so the exact fallback answers them (round 2: such rows came back
    poisoned from ``search_device`` and were merged silently; now the call answers them itself)."""
    corpus = oracle.seeded_unit_rows(n, 384, 1234)
    queries = oracle.seeded_unit_rows(nq, 384, 4321)
    if dups:
        corpus[1000:1320] = corpus[5]
        corpus[n // 2 + 700 : n // 2 + 1020] = corpus[5]
        queries = np.repeat(corpus[5][None], nq, axis=0) + 0.01 * queries
        queries = (queries / np.linalg.norm(queries, axis=1, keepdims=True)).astype(np.float32)
    return corpus, queries


def _gpu_worker(rank, world, port, n, nq, k, out_dir, backend, dups=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semantic_search_kd_amd import FAISSIndexBuilder

        corpus, queries = _sharded_case(n, nq, dups)
        queries = torch.from_numpy(queries).to(dev)
        lo, hi = shard_bounds(n, world, rank)
        index = FAISSIndexBuilder(embedding_dim=384, metric="ip", device=str(dev), id_offset=lo)
        index.add(corpus[lo:hi])

        def local_search(q, kk, out_scores=None, out_ids=None):
            return index.search_device(q, kk, normalize_queries=False, out_scores=out_scores, out_ids=out_ids)

        def host_staged_gather(out, inp):  # one GPU shared by both ranks: gloo moves host copies
            host_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host_out, inp.cpu())
            out.copy_(host_out)

        searcher = ShardedSearcher(local_search, all_gather=None if backend == "nccl" else host_staged_gather)
        s, i = searcher.search(queries, k)
        torch.cuda.synchronize()
        assert searcher.last_world == world
        fallback = -1 if index.last_status is None else int(index.last_status[1])
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), s=s.cpu().numpy(), i=i.cpu().numpy(), fallback=fallback)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n,nq,k,dups", [(5000, 70, 10, False), (5, 3, 10, False), (8192, 1200, 10, True)])
def test_product_sharded_search_two_ranks(tmp_path, n, nq, k, dups):
    """search_device -> packed record -> all-gather -> sskd_topk_merge_packed in 2 processes; the third case
    drives > 1 024 queries per rank through the screened search's in-call exact fallback (duplicate-heavy
    shards): the merged result must still be the oracle's, bit for bit."""
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    world = 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, _PORTS.setdefault((n, nq), _free_port()), n, nq, k, str(tmp_path), backend, dups))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    corpus, queries = _sharded_case(n, nq, dups)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, k)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["i"], ref_i) and np.array_equal(got["s"], ref_s)
        if dups:
            assert int(got["fallback"]) > 1024, int(got["fallback"])   # the regime that used to poison rows


_PORTS = {}
