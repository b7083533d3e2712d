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
other ranks sit in ``serve_forever()``.  One search = broadcast of a 3-word header and the query block from rank 0,
then ``dist.ShardedSearcher`` (local scan -> one all-gather -> merge).  ``torch.distributed`` backend ``nccl`` is
RCCL over xGMI; ``gloo`` works for rehearsals (records are then staged through the host).
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
                 device: Optional[str] = None, group=None, index_factory: Optional[Callable] = None) -> None:
        self.embedding_dim, self.index_type, self.metric, self.device, self.group = embedding_dim, index_type, metric, device, group
        self._factory = index_factory or _default_factory
        self.local = None                      # this rank's shard(s): a FAISSIndexBuilder
        self.doc_ids: List[str] = []           # rank 0: every shard's ids in global row order
        self.doc_texts: Optional[Dict[str, str]] = None
        self.ntotal = 0
        self.manifest: Optional[Dict] = None
        self._searcher: Optional[ShardedSearcher] = None

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

    def _header(self, op: int = 0, a: int = 0, b: int = 0) -> Tuple[int, int, int]:
        """rank 0 announces the next collective step; the others learn it"""
        import torch.distributed as dist

        world, rank = _world(self.group)
        if world == 1:
            return op, a, b
        h = torch.tensor([op, a, b], dtype=torch.int64, device=self._comm_device())
        dist.broadcast(h, src=self._src(), group=self.group)
        return int(h[0]), int(h[1]), int(h[2])

    # ------------------------------------------------------------------ load
    def _load_local(self, index_dir: Path) -> None:
        world, rank = _world(self.group)
        manifest = json.loads((index_dir / MANIFEST).read_text())
        if manifest.get("embedding_dim", self.embedding_dim) != self.embedding_dim:
            raise ValueError(f"index has dim {manifest['embedding_dim']}, expected {self.embedding_dim}")
        shards = manifest["shards"]
        s = len(shards)
        mine = shards[rank * s // world : (rank + 1) * s // world]   # a contiguous run (possibly empty)
        offset = mine[0]["id_offset"] if mine else manifest["n_total"]
        local = self._factory(self.embedding_dim, manifest.get("metric", self.metric), self.device, offset)
        for j, sh in enumerate(mine):
            local.load(index_dir / sh["dir"], append=j > 0)
        if not mine:
            local.id_offset = offset
        self.local, self.manifest, self.ntotal = local, manifest, int(manifest["n_total"])
        self.metric = manifest.get("metric", self.metric)
        if rank == 0:   # the serving rank maps global row ids to doc ids / texts
            ids: List[str] = []
            texts: Dict[str, str] = {}
            for sh in shards:
                ids.extend(json.loads((index_dir / sh["dir"] / "doc_ids.json").read_text()))
                tp = index_dir / sh["dir"] / "texts.json"
                if tp.exists():
                    texts.update(json.loads(tp.read_text()))
            if len(ids) != self.ntotal:
                raise ValueError(f"{index_dir}: {len(ids)} doc ids for {self.ntotal} rows")
            self.doc_ids, self.doc_texts = ids, (texts or None)
        self._searcher = ShardedSearcher(self._local_search, group=self.group)

    def load(self, index_dir: Union[str, Path]) -> None:
        """Collective: on rank 0 (the caller in a served deployment) this tells the ranks waiting in
        ``serve_forever`` to load the same directory.  When every rank calls it directly (start-up), pass through
        ``load_all_ranks`` instead."""
        import torch.distributed as dist

        index_dir = Path(index_dir)
        world, rank = _world(self.group)
        if world > 1:
            if rank != 0:
                raise RuntimeError("ShardedIndex.load is rank 0's call; the other ranks run serve_forever()")
            self._header(_OP_LOAD)
            dist.broadcast_object_list([str(index_dir)], src=self._src(), group=self.group)
        self._load_local(index_dir)

    def load_all_ranks(self, index_dir: Union[str, Path]) -> None:
        """Start-up form: EVERY rank calls this with the same directory (no announcement needed)."""
        self._load_local(Path(index_dir))

    # ------------------------------------------------------------------ search
    def _local_search(self, queries: torch.Tensor, k: int, out_scores=None, out_ids=None):
        return self.local.search_device(queries, k, normalize_queries=None, out_scores=out_scores, out_ids=out_ids)

    def _collective_search(self, queries: Optional[torch.Tensor], nq: int, k: int):
        import torch.distributed as dist

        world, rank = _world(self.group)
        dev = torch.device(self.local.device)
        if world > 1:
            comm = self._comm_device()
            buf = queries.to(comm) if rank == 0 else torch.empty((nq, self.embedding_dim), dtype=torch.float32, device=comm)
            dist.broadcast(buf, src=self._src(), group=self.group)
            queries = buf
        return self._searcher.search(queries.to(dev), k)

    def search(self, query_emb: np.ndarray, k: int = 10) -> Tuple[np.ndarray, np.ndarray]:
        """``(distances [nq, k] fp32 desc, GLOBAL row ids [nq, k] int64, -1 padded)`` - rank 0's call."""
        if self.local is None:
            raise RuntimeError("index is empty: call load first")
        world, rank = _world(self.group)
        if world > 1 and rank != 0:
            raise RuntimeError("ShardedIndex.search is rank 0's call; the other ranks run serve_forever()")
        q = np.ascontiguousarray(np.asarray(query_emb, dtype=np.float32))
        if q.ndim == 1:
            q = q[None, :]
        self._header(_OP_SEARCH, q.shape[0], k)
        s, i = self._collective_search(torch.from_numpy(q), q.shape[0], k)
        return s.cpu().numpy(), i.cpu().numpy()

    def serve_forever(self) -> None:
        """Ranks other than 0: answer rank 0's announcements until it says stop."""
        import torch.distributed as dist

        while True:
            op, a, b = self._header()
            if op == _OP_STOP:
                return
            if op == _OP_SEARCH:
                self._collective_search(None, a, b)
            elif op == _OP_LOAD:
                box = [None]
                dist.broadcast_object_list(box, src=self._src(), group=self.group)
                self._load_local(Path(box[0]))

    def close(self) -> None:
        """rank 0: release the ranks waiting in ``serve_forever``"""
        world, rank = _world(self.group)
        if world > 1 and rank == 0:
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
