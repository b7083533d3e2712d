"""Row-sharded index: build, persist, reload and serve across the GPUs of one node.

The reference builds ONE in-process index (scripts/build_faiss_index.py:45-72) and serves it from ONE process
(src/serve/app.py:407-457).  BASELINE.json's north star shards the corpus row-wise over 8 GPUs (SURVEY.md section
8e): "each GPU writes its embeddings straight into its own index shard" - zero communication while building - and a
search is: queries replicated, every rank scans its shard, ONE all-gather of the packed per-shard top-k, merge.

On disk (``<output_dir>/``)::

    shards.json                 {"format": 1, "n_total": N, "embedding_dim": 384, "metric": "cosine",
                                 "shards": [{"dir": "shard_0", "id_offset": 0, "rows": n0}, ...]}
    shard_<r>/index.faiss       this shard's vectors (flat inner-product layout, as an unsharded index)
    shard_<r>/doc_ids.json      this shard's ids, in row order
    shard_<r>/texts.json        this shard's id -> text table
    shard_<r>/shard.json        {"id_offset", "rows", "rank", "world_size", "n_total"}

Every shard directory is by itself a valid ``FAISSIndexBuilder.load`` target.  An index written by G ranks can be
served by any world size that divides the work into contiguous runs of shards (8 shards on 1, 2, 4 or 8 GPUs): rank r
loads shards ``[r S / G, (r + 1) S / G)`` back to back into one HBM buffer.

Serving (``ShardedIndex``): rank 0 owns the HTTP surface and calls ``search`` like on a ``FAISSIndexBuilder``; the
other ranks sit in ``serve_forever()``.  One search = a 3-word header on the CONTROL group, the query block from
rank 0, every rank's local scan, one status word per rank on the control group, then - only when every rank
succeeded - ``dist.ShardedSearcher``'s all-gather of the packed records and the merge.  ``torch.distributed`` backend
``nccl`` is RCCL over xGMI; ``gloo`` works for rehearsals (records are then staged through the host).

Failure path (reference: src/serve/app.py:354-361 turns any exception into a 500; SURVEY.md section 5: a failed
HIP / RCCL call must surface as a Python exception and ``/health.index_loaded`` must reflect shard state):

* the control group is a CPU (``gloo``) group of the same ranks with an effectively unbounded timeout: the waiting
  ranks park in a HOST broadcast, never inside a device collective (an RCCL collective that rank 0 has not joined
  is aborted by the watchdog after the group's timeout - an idle server would die - and spins a GPU meanwhile);
* rank 0 validates arguments and the manifest BEFORE it announces an operation; every rank runs its part of the
  operation inside ``try`` and reports one status word; the device collective of the data path starts only when all
  words are zero.  Otherwise every rank skips it, rank 0 raises ``ShardFailure`` (-> the route's 500) naming the
  ranks and their messages, and the deployment keeps serving: a failed ``load`` leaves the previous index in place
  on EVERY rank (two-phase: prepare, exchange, commit);
* status exchanges wait ``op_timeout_s``: a rank that died or hangs turns into ``ShardFailure`` on rank 0 within that
  time, the index is marked broken (``is_loaded`` False -> ``/health.index_loaded`` False) and later calls fail fast.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .dist import ShardedSearcher, shard_bounds

MANIFEST = "shards.json"
_OP_STOP, _OP_SEARCH, _OP_LOAD = 0, 1, 2


class ShardFailure(RuntimeError):
    """An operation of the sharded deployment failed on one or more ranks (or a rank did not answer in time)."""

    def __init__(self, what: str, failures: Dict[int, str]) -> None:
        self.failures = dict(failures)
        detail = "; ".join(f"rank {r}: {m}" for r, m in sorted(self.failures.items()))
        super().__init__(f"{what} failed on {len(self.failures)} rank(s): {detail}")


def _world(group=None) -> Tuple[int, int]:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _default_factory(embedding_dim: int, metric: str, device, id_offset: int):
    from .index import FAISSIndexBuilder

    return FAISSIndexBuilder(embedding_dim=embedding_dim, index_type="HNSW", metric=metric, device=device, id_offset=id_offset)


def build_sharded(model, parquet_path: Union[str, Path], output_dir: Union[str, Path], batch_size: int = 32,
                  max_docs: Optional[int] = None, device: Optional[str] = None, group=None, embedding_dim: int = 384,
                  metric: str = "cosine", index_factory: Optional[Callable] = None, show_progress: bool = False,
                  text_column: str = "text", id_column: str = "chunk_id") -> Dict:
    """Every rank encodes ITS contiguous row range ``shard_bounds(N, G, r)`` of the corpus straight into its own HBM
    shard (``id_offset`` = first row) and saves it as ``shard_<r>/``; rank 0 writes the manifest.  No data-path
    communication - two barriers order the directory creation and the manifest.  Works without a process group
    (one shard).  Returns the manifest."""
    import torch.distributed as dist

    from .index import read_corpus_parquet

    world, rank = _world(group)
    out = Path(output_dir)
    ids, texts = read_corpus_parquet(parquet_path, max_docs, text_column, id_column)
    n = len(texts)
    lo, hi = shard_bounds(n, world, rank)
    factory = index_factory or _default_factory
    builder = factory(embedding_dim, metric, device, lo)
    builder.reserve(max(hi - lo, 1))
    slab = max(batch_size, 65536)   # stream: a multi-million-passage shard never needs one host matrix
    on_device = getattr(model, "encode_documents_device", None)   # embeddings go encoder -> index tiles inside HBM
    for a in range(lo, hi, slab):
        b = min(a + slab, hi)
        if on_device is not None:
            builder.add(on_device(texts[a:b], batch_size=batch_size))
        else:
            builder.add(model.encode_documents(texts[a:b], batch_size=batch_size, show_progress=show_progress))
    builder.doc_ids = ids[lo:hi]
    builder.doc_texts = dict(zip(ids[lo:hi], texts[lo:hi]))
    builder.shard_info = {"rank": rank, "world_size": world, "n_total": n}
    if rank == 0:
        out.mkdir(parents=True, exist_ok=True)
    if world > 1:
        dist.barrier(group)
    builder.save(out / f"shard_{rank}")
    if world > 1:
        dist.barrier(group)
    manifest = {
        "format": 1, "n_total": n, "embedding_dim": embedding_dim, "metric": metric,
        "shards": [{"dir": f"shard_{r}", "id_offset": shard_bounds(n, world, r)[0],
                    "rows": shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0]} for r in range(world)],
    }
    if rank == 0:
        (out / MANIFEST).write_text(json.dumps(manifest, indent=1) + "\n")
    if world > 1:
        dist.barrier(group)
    return manifest


def is_sharded_dir(index_dir: Union[str, Path]) -> bool:
    return (Path(index_dir) / MANIFEST).exists()


class ShardedIndex:
    """``FAISSIndexBuilder``-shaped front of a row-sharded index (``search`` / ``load`` / ``doc_ids`` / ``ntotal``)."""

    def __init__(self, embedding_dim: int = 384, index_type: str = "HNSW", metric: str = "cosine",
                 device: Optional[str] = None, group=None, index_factory: Optional[Callable] = None,
                 ctrl_group=None, op_timeout_s: float = 60.0, idle_timeout_s: float = 365 * 86400.0) -> None:
        """``group``: the data-path group (queries, packed records); ``ctrl_group``: a gloo group of the same ranks for
        headers and status words (made on first use when ``group`` is the default group); ``op_timeout_s``: how long a
        status exchange may take before the missing ranks are declared failed; ``idle_timeout_s``: how long the
        waiting ranks may sit without an announcement from rank 0."""
        self.embedding_dim, self.index_type, self.metric, self.device, self.group = embedding_dim, index_type, metric, device, group
        self._factory = index_factory or _default_factory
        self.local = None                      # this rank's shard(s): a FAISSIndexBuilder
        self.doc_ids: List[str] = []           # rank 0: every shard's ids in global row order
        self.doc_texts: Optional[Dict[str, str]] = None
        self.ntotal = 0
        self.manifest: Optional[Dict] = None
        self._searcher: Optional[ShardedSearcher] = None
        self._ctrl_group = ctrl_group
        self.op_timeout_s, self.idle_timeout_s = float(op_timeout_s), float(idle_timeout_s)
        self.broken: Optional[str] = None      # set when a rank stopped answering: the process group is unusable
        self.last_failure: Optional[Dict[int, str]] = None

    # ------------------------------------------------------------------ collective plumbing
    def _comm_device(self) -> torch.device:
        import torch.distributed as dist

        world, _ = _world(self.group)
        if world > 1 and dist.get_backend(self.group) != "gloo":
            return torch.device(self.local.device if self.local is not None else (self.device or "cuda"))
        return torch.device("cpu")

    def _src(self) -> int:
        import torch.distributed as dist

        return dist.get_global_rank(self.group, 0) if self.group is not None else 0

    def _ctrl(self):
        """The control group: CPU tensors over gloo, made collectively on first use (every rank's first use is the
        same step: ``load_all_ranks`` at start-up, or the first header)."""
        import datetime

        import torch.distributed as dist

        if self._ctrl_group is None:
            if self.group is not None:
                if dist.get_backend(self.group) != "gloo":
                    raise RuntimeError("ShardedIndex over a sub-group needs ctrl_group= (a gloo group of the same ranks)")
                self._ctrl_group = self.group
            else:
                self._ctrl_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=self.idle_timeout_s))
        return self._ctrl_group

    def _header(self, op: int = 0, a: int = 0, b: int = 0) -> Tuple[int, int, int]:
        """rank 0 announces the next collective step; the others learn it (a HOST broadcast: waiting costs no GPU
        and no watchdog can abort it)"""
        import torch.distributed as dist

        world, rank = _world(self.group)
        if world == 1:
            return op, a, b
        h = torch.tensor([op, a, b], dtype=torch.int64)
        dist.broadcast(h, src=self._src(), group=self._ctrl())
        return int(h[0]), int(h[1]), int(h[2])

    def _exchange_status(self, what: str, error: Optional[BaseException]) -> Dict[int, str]:
        """Every rank reports one word (0 = its part succeeded); returns ``{rank: message}`` of the failed ranks, the
        same on every rank.  A rank that does not answer within ``op_timeout_s`` breaks the deployment."""
        import datetime

        import torch.distributed as dist

        world, rank = _world(self.group)
        if world == 1:
            return {0: f"{type(error).__name__}: {error}"} if error is not None else {}
        ctrl = self._ctrl()
        mine = torch.tensor([0 if error is None else 1], dtype=torch.int64)
        words = torch.zeros(world, dtype=torch.int64)
        try:
            work = dist.all_gather_into_tensor(words, mine, group=ctrl, async_op=True)
            work.wait(datetime.timedelta(seconds=self.op_timeout_s))
        except Exception as exc:  # noqa: BLE001 - a peer died or hangs: nothing collective can be trusted any more
            self.broken = f"{what}: a rank did not report within {self.op_timeout_s:.0f} s ({type(exc).__name__}: {exc})"
            raise ShardFailure(what, {-1: self.broken}) from exc
        failed = [r for r in range(world) if int(words[r]) != 0]
        if not failed:
            return {}
        # rare path: collect the messages (every rank saw the same words, so every rank joins)
        texts: List[Optional[str]] = [None] * world
        dist.all_gather_object(texts, None if error is None else f"{type(error).__name__}: {error}", group=ctrl)
        return {r: texts[r] or "failed" for r in failed}

    # ------------------------------------------------------------------ load
    @staticmethod
    def read_manifest(index_dir: Path, embedding_dim: int) -> Dict:
        """parse and validate ``shards.json`` (rank 0 does this BEFORE it announces a load)"""
        manifest = json.loads((Path(index_dir) / MANIFEST).read_text())
        if manifest.get("embedding_dim", embedding_dim) != embedding_dim:
            raise ValueError(f"index has dim {manifest['embedding_dim']}, expected {embedding_dim}")
        shards = manifest.get("shards")
        if not isinstance(shards, list) or not shards or "n_total" not in manifest:
            raise ValueError(f"{index_dir}/{MANIFEST}: no shards listed")
        at = 0
        for sh in shards:
            if int(sh["id_offset"]) != at:
                raise ValueError(f"{index_dir}/{MANIFEST}: shard {sh['dir']} starts at row {sh['id_offset']}, expected {at}")
            at += int(sh["rows"])
        if at != int(manifest["n_total"]):
            raise ValueError(f"{index_dir}/{MANIFEST}: shards hold {at} rows, n_total says {manifest['n_total']}")
        return manifest

    def _prepare_local(self, index_dir: Path) -> Dict:
        """phase 1 of a load: this rank's shards into a NEW local index; nothing of the serving state is touched"""
        world, rank = _world(self.group)
        manifest = self.read_manifest(index_dir, self.embedding_dim)
        shards = manifest["shards"]
        s = len(shards)
        mine = shards[rank * s // world : (rank + 1) * s // world]   # a contiguous run (possibly empty)
        offset = mine[0]["id_offset"] if mine else manifest["n_total"]
        local = self._factory(self.embedding_dim, manifest.get("metric", self.metric), self.device, offset)
        for j, sh in enumerate(mine):
            local.load(index_dir / sh["dir"], append=j > 0)
        if not mine:
            local.id_offset = offset
        rows = sum(int(sh["rows"]) for sh in mine)
        have = getattr(local, "ntotal", None)
        if have is not None and int(have) != rows:
            raise ValueError(f"{index_dir}: rank {rank} loaded {have} rows, the manifest lists {rows}")
        staged = {"local": local, "manifest": manifest, "doc_ids": [], "doc_texts": None}
        if rank == 0:   # the serving rank maps global row ids to doc ids / texts
            ids: List[str] = []
            texts: Dict[str, str] = {}
            for sh in shards:
                ids.extend(json.loads((index_dir / sh["dir"] / "doc_ids.json").read_text()))
                tp = index_dir / sh["dir"] / "texts.json"
                if tp.exists():
                    texts.update(json.loads(tp.read_text()))
            if len(ids) != int(manifest["n_total"]):
                raise ValueError(f"{index_dir}: {len(ids)} doc ids for {manifest['n_total']} rows")
            staged["doc_ids"], staged["doc_texts"] = ids, (texts or None)
        return staged

    def _commit(self, staged: Dict) -> None:
        old = self.local
        self.local, self.manifest = staged["local"], staged["manifest"]
        self.ntotal = int(self.manifest["n_total"])
        self.metric = self.manifest.get("metric", self.metric)
        self.doc_ids, self.doc_texts = staged["doc_ids"], staged["doc_texts"]
        self._searcher = ShardedSearcher(self._local_search, group=self.group)
        if old is not None and old is not self.local and hasattr(old, "cleanup"):
            old.cleanup()

    def _load_local(self, index_dir: Path) -> None:
        """prepare -> one status word per rank -> commit on EVERY rank or on none"""
        staged, error = None, None
        try:
            staged = self._prepare_local(Path(index_dir))
        except Exception as exc:  # noqa: BLE001 - reported to every rank below
            error = exc
        failures = self._exchange_status(f"load {index_dir}", error)
        self.last_failure = failures or None
        if failures:
            if staged is not None and hasattr(staged["local"], "cleanup"):
                staged["local"].cleanup()
            raise ShardFailure(f"load {index_dir}", failures) from error
        self._commit(staged)

    def load(self, index_dir: Union[str, Path]) -> None:
        """Collective: on rank 0 (the caller in a served deployment) this tells the ranks waiting in
        ``serve_forever`` to load the same directory.  When every rank calls it directly (start-up), pass through
        ``load_all_ranks`` instead.  Raises ``ShardFailure`` when any rank cannot load its shards; the index that
        was being served stays in place on every rank."""
        import torch.distributed as dist

        index_dir = Path(index_dir)
        world, rank = _world(self.group)
        if world > 1:
            if rank != 0:
                raise RuntimeError("ShardedIndex.load is rank 0's call; the other ranks run serve_forever()")
            self._check_usable()
            self.read_manifest(index_dir, self.embedding_dim)   # nothing is announced for a directory rank 0 cannot read
            self._header(_OP_LOAD)
            dist.broadcast_object_list([str(index_dir)], src=self._src(), group=self._ctrl())
        self._load_local(index_dir)

    def load_all_ranks(self, index_dir: Union[str, Path]) -> None:
        """Start-up form: EVERY rank calls this with the same directory (no announcement needed).  Raises
        ``ShardFailure`` on every rank when any rank fails, so that the launcher exits instead of hanging."""
        self._load_local(Path(index_dir))

    # ------------------------------------------------------------------ state
    @property
    def is_loaded(self) -> bool:
        """every rank holds its shards of the index being served and answers (``/health.index_loaded``)"""
        return self.local is not None and self.broken is None

    def health(self) -> Dict:
        world, _ = _world(self.group)
        return {"world_size": world, "loaded": self.is_loaded, "ntotal": self.ntotal,
                "shards": len(self.manifest["shards"]) if self.manifest else 0,
                "broken": self.broken, "last_failure": self.last_failure}

    def _check_usable(self) -> None:
        if self.broken is not None:
            raise ShardFailure("sharded index", {-1: f"deployment is broken ({self.broken}); restart it"})

    # ------------------------------------------------------------------ search
    def _local_search(self, queries: torch.Tensor, k: int, out_scores=None, out_ids=None):
        return self.local.search_device(queries, k, normalize_queries=None, out_scores=out_scores, out_ids=out_ids)

    def _collective_search(self, queries: Optional[torch.Tensor], nq: int, k: int):
        import torch.distributed as dist

        world, rank = _world(self.group)
        if world == 1:
            return self._searcher.search(queries.to(torch.device(self.local.device)), k)
        error, partial = None, None
        comm = self._comm_device()
        buf = queries.to(comm) if rank == 0 else torch.empty((nq, self.embedding_dim), dtype=torch.float32, device=comm)
        dist.broadcast(buf, src=self._src(), group=self.group)
        try:
            if self.local is None:
                raise RuntimeError("no index loaded on this rank")
            partial = self._searcher.search_local(buf.to(torch.device(self.local.device)), k)
        except Exception as exc:  # noqa: BLE001 - reported below; this rank still answers the status exchange
            error = exc
        failures = self._exchange_status("search", error)
        self.last_failure = failures or None
        if failures:
            if rank == 0:
                raise ShardFailure("search", failures) from error
            return None
        return self._searcher.gather_merge(partial, nq, k)

    def search(self, query_emb: np.ndarray, k: int = 10) -> Tuple[np.ndarray, np.ndarray]:
        """``(distances [nq, k] fp32 desc, GLOBAL row ids [nq, k] int64, -1 padded)`` - rank 0's call."""
        if self.local is None:
            raise RuntimeError("index is empty: call load first")
        world, rank = _world(self.group)
        if world > 1 and rank != 0:
            raise RuntimeError("ShardedIndex.search is rank 0's call; the other ranks run serve_forever()")
        self._check_usable()
        # everything that can be wrong with the CALL is found before the other ranks hear of it
        q = np.ascontiguousarray(np.asarray(query_emb, dtype=np.float32))
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.embedding_dim:
            raise ValueError(f"queries have shape {q.shape}, expected [nq, {self.embedding_dim}]")
        if int(k) < 1:
            raise ValueError(f"k must be >= 1, got {k}")
        if q.shape[0] == 0:
            return np.zeros((0, int(k)), np.float32), np.zeros((0, int(k)), np.int64)
        self._header(_OP_SEARCH, q.shape[0], int(k))
        s, i = self._collective_search(torch.from_numpy(q), q.shape[0], int(k))
        return s.cpu().numpy(), i.cpu().numpy()

    def serve_forever(self) -> None:
        """Ranks other than 0: answer rank 0's announcements until it says stop.  An operation that fails HERE is
        reported to rank 0 through the status word and the loop goes on; only a broken process group ends it."""
        import torch.distributed as dist

        while True:
            op, a, b = self._header()
            if op == _OP_STOP:
                return
            try:
                if op == _OP_SEARCH:
                    self._collective_search(None, a, b)
                elif op == _OP_LOAD:
                    box = [None]
                    dist.broadcast_object_list(box, src=self._src(), group=self._ctrl())
                    self._load_local(Path(box[0]))
            except ShardFailure:
                if self.broken is not None:
                    raise
                # some rank's part failed (maybe this one's): rank 0 has raised it to its caller; keep serving

    def close(self) -> None:
        """rank 0: release the ranks waiting in ``serve_forever``"""
        world, rank = _world(self.group)
        if world > 1 and rank == 0 and self.broken is None:
            self._header(_OP_STOP)

    def cleanup(self) -> None:
        if self.local is not None:
            self.local.cleanup()


def open_index(index_dir: Union[str, Path], embedding_dim: int = 384, current=None, device: Optional[str] = None):
    """What the serving layer calls for ``/index/load`` (reference: src/serve/app.py:407-441): a directory with a
    ``shards.json`` manifest opens as a ``ShardedIndex`` (re-using ``current`` when it already is one, so that the
    waiting ranks follow), anything else as a plain ``FAISSIndexBuilder``."""
    from .index import FAISSIndexBuilder

    index_dir = Path(index_dir)
    if is_sharded_dir(index_dir):
        idx = current if isinstance(current, ShardedIndex) else ShardedIndex(embedding_dim=embedding_dim, device=device)
        idx.load(index_dir)
        return idx
    if isinstance(current, ShardedIndex) and _world(current.group)[0] > 1:
        raise ValueError(f"{index_dir} has no {MANIFEST}: a {_world(current.group)[0]}-rank deployment serves sharded indexes")
    builder = FAISSIndexBuilder(embedding_dim=embedding_dim, device=device)
    builder.load(index_dir)
    return builder
