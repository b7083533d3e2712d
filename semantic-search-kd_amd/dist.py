"""Row-sharded search across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has a single in-process index (src/serve/app.py:49-66); BASELINE.json's
north star shards the corpus row-wise, so this is the one place the path has a real
exchange step:

    every rank: scan its shard  ->  one packed record { ids int64[nq, k]; scores fp32[nq, k] }
                                    (the scan's final merge writes straight into the record)
    all ranks : ONE all-gather  ->  [G] records             (1.2 MB / rank at nq=10k, k=10)
    every rank: merge G*k -> k per query (``sskd_topk_merge_packed``; ties: lower global id)

The collective goes through ``torch.distributed`` (backend ``nccl`` = RCCL on ROCm) on
the same stream as the kernels.  ``local_search`` / ``merge`` / ``all_gather`` are injectable
so that the sharding logic can be exercised with ``gloo`` on CPU in the tests.
"""
from __future__ import annotations

import inspect
from typing import Callable, Optional, Tuple

import torch

from . import _native


def shard_bounds(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range ``[lo, hi)`` of ``rank``: ceil(N / G) rows per rank."""
    per = -(-n_rows // world_size)
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


def record_bytes(nq: int, k: int) -> int:
    """Bytes of one rank's packed record (``sskd_topk_record_bytes``): ids, scores, pad to 16."""
    return (nq * k * 12 + 15) // 16 * 16 if nq > 0 and k > 0 else 0


def record_views(record: torch.Tensor, nq: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(scores fp32 [nq, k], ids int64 [nq, k])`` views into a packed uint8 record."""
    ids = record[: nq * k * 8].view(torch.int64).view(nq, k)
    scores = record[nq * k * 8 : nq * k * 12].view(torch.float32).view(nq, k)
    return scores, ids


def hip_merge(scores: torch.Tensor, ids: torch.Tensor, k_out: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``[G, nq, k_in]`` device lists -> ``[nq, k_out]`` via the HIP merge kernel."""
    lib = _native.load()
    g, nq, k_in = scores.shape
    out_s = torch.empty((nq, k_out), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k_out), dtype=torch.int64, device=scores.device)
    _native.check(
        lib.sskd_topk_merge(
            scores.contiguous().data_ptr(),
            ids.contiguous().data_ptr(),
            g,
            nq,
            k_in,
            k_out,
            out_s.data_ptr(),
            out_i.data_ptr(),
            int(torch.cuda.current_stream(scores.device).cuda_stream),
        )
    )
    return out_s, out_i


def hip_merge_packed(records: torch.Tensor, g: int, nq: int, k_in: int, k_out: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``[G * record_bytes]`` gathered uint8 records -> ``[nq, k_out]`` (no unpacking copies)."""
    lib = _native.load()
    out_s = torch.empty((nq, k_out), dtype=torch.float32, device=records.device)
    out_i = torch.empty((nq, k_out), dtype=torch.int64, device=records.device)
    _native.check(
        lib.sskd_topk_merge_packed(
            records.data_ptr(), g, nq, k_in, k_out, out_s.data_ptr(), out_i.data_ptr(),
            int(torch.cuda.current_stream(records.device).cuda_stream),
        )
    )
    return out_s, out_i


def sharded_scores(score_fn: Callable[[int, int], torch.Tensor], n_items: int, group=None,
                   device: Optional[torch.device] = None) -> torch.Tensor:
    """Pure data parallelism over ``n_items`` independent work items (teacher scoring, BASELINE cfg 5;
    SURVEY.md section 8e: "pure DP over pairs with a final concat").

    Rank r computes ``score_fn(lo, hi) -> fp32 [hi - lo]`` for its contiguous range ``shard_bounds(n_items, G, r)``;
    ONE ``all_gather_into_tensor`` of ceil(n / G) floats per rank is the final concat - every rank returns all
    ``n_items`` scores in item order.  No collective inside the scoring itself.  Without an initialised process
    group (or world size 1) this is just ``score_fn(0, n_items)``."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return score_fn(0, n_items)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n_items, world, rank)
    local = score_fn(lo, hi) if hi > lo else torch.empty(0, dtype=torch.float32, device=device)
    # gloo moves host tensors, nccl (= RCCL) device tensors
    comm_dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else local.device
    per = -(-n_items // world)
    send = torch.zeros(per, dtype=torch.float32, device=comm_dev)
    send[: hi - lo] = local.to(comm_dev, torch.float32)
    recv = torch.empty(world * per, dtype=torch.float32, device=comm_dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv[:n_items].to(local.device)   # ceil-sized shards: only the tail of the LAST ranks is padding


class ShardedSearcher:
    """Search a corpus whose rows are split over the ranks of a process group."""

    def __init__(
        self,
        local_search: Callable[..., Tuple[torch.Tensor, torch.Tensor]],
        group=None,
        merge: Optional[Callable[[torch.Tensor, torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]]] = None,
        all_gather: Optional[Callable[[torch.Tensor, torch.Tensor], None]] = None,
    ) -> None:
        """``local_search(queries, k[, out_scores=, out_ids=])`` must return this rank's
        ``(scores, GLOBAL ids)`` (e.g. ``FAISSIndexBuilder(id_offset=lo).search_device``); when it
        accepts ``out_scores`` / ``out_ids`` it writes straight into the packed record.
        ``merge(scores [G, nq, k], ids [G, nq, k], k)`` replaces the HIP merge (CPU tests);
        ``all_gather(out, inp)`` replaces ``dist.all_gather_into_tensor`` (e.g. host-staged gloo
        when several test ranks share one GPU)."""
        self.local_search = local_search
        self.group = group
        self.merge = merge
        self.all_gather = all_gather
        self.last_world = 1

    def _local_into(self, queries, k, out_s, out_i):
        fn = self.local_search
        try:
            params = inspect.signature(fn).parameters
        except (TypeError, ValueError):
            params = {}
        if "out_scores" in params and "out_ids" in params:
            s, i = fn(queries, k, out_scores=out_s, out_ids=out_i)
        else:
            s, i = fn(queries, k)
        if s.data_ptr() != out_s.data_ptr():
            out_s.copy_(s)
        if i.data_ptr() != out_i.data_ptr():
            out_i.copy_(i)

    def search_local(self, queries: torch.Tensor, k: int) -> torch.Tensor:
        """This rank's half of a sharded search, no communication: the local scan writes its ``(scores, GLOBAL
        ids)`` straight into a fresh packed record, which is returned (``gather_merge`` takes it from there).
        Split from ``search`` so that a caller can agree with the other ranks that every local scan succeeded
        BEFORE any rank enters the device collective (sharded_index.ShardedIndex)."""
        nq = queries.shape[0]
        send = torch.empty(record_bytes(nq, k), dtype=torch.uint8, device=queries.device)
        out_s, out_i = record_views(send, nq, k)
        self._local_into(queries, k, out_s, out_i)
        return send

    def gather_merge(self, send: torch.Tensor, nq: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """ONE all-gather of the packed records (RCCL over xGMI on a GPU node) and the merge of the G lists."""
        import torch.distributed as dist

        world = dist.get_world_size(self.group)
        self.last_world = world
        rec = record_bytes(nq, k)
        dev = send.device
        recv = torch.empty(world * rec, dtype=torch.uint8, device=dev)
        if self.all_gather is not None:
            self.all_gather(recv, send)
        elif dist.get_backend(self.group) == "gloo" and send.is_cuda:
            # gloo moves host memory (several ranks sharing one GPU in tests, CPU-only rehearsals): stage the
            # 12-byte-per-result records through the host.  On a node with one GPU per rank the backend is nccl
            # (= RCCL over xGMI) and the records never leave HBM.
            host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(host, send.cpu(), group=self.group)
            recv.copy_(host)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        if self.merge is not None:
            table = recv.view(world, rec)
            all_i = table[:, : nq * k * 8].contiguous().view(torch.int64).view(world, nq, k)
            all_s = table[:, nq * k * 8 : nq * k * 12].contiguous().view(torch.float32).view(world, nq, k)
            return self.merge(all_s, all_i, k)
        return hip_merge_packed(recv, world, nq, k, k)

    def search(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            self.last_world = 1
            return self.local_search(queries, k)
        nq = queries.shape[0]
        if record_bytes(nq, k) == 0:
            return self.local_search(queries, k)
        return self.gather_merge(self.search_local(queries, k), nq, k)
