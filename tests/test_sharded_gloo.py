"""Row-sharded search over world_size-2 ``gloo`` process groups on CPU.

Exercises the N > 1 plumbing of ``ShardedSearcher`` — shard bounds, global id offsets, the
all-gather layout ``[G, nq, k]`` and the merge — with the HIP calls replaced by the oracle
(tests may use the oracle; the product default is ``hip_merge`` / ``search_device``).
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


@pytest.mark.parametrize("n,nq,k", [(1000, 17, 10), (5, 3, 10)])
def test_sharded_search_equals_unsharded(tmp_path, n, nq, k):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, nq, k, str(tmp_path)), nprocs=world, join=True)
    corpus = oracle.seeded_unit_rows(n, 384, 1234)
    queries = oracle.seeded_unit_rows(nq, 384, 4321)
    ref_s, ref_i = oracle.topk_fma(queries, corpus, k)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["i"], ref_i) and np.array_equal(got["s"], ref_s)
