"""``FAISSIndexBuilder``-shaped index backed by the gfx950 exact top-k scan.

Mirrors the (absent-from-checkout) reference class ``src.index.build_index.FAISSIndexBuilder``
as reconstructed from its call sites:

* ctor ``(embedding_dim=384, index_type="HNSW", metric="cosine")`` — scripts/build_faiss_index.py:49-53,
  src/serve/app.py:427-429
* ``build_from_parquet(model=, parquet_path=, batch_size=, max_docs=, hnsw_m=, hnsw_ef_construction=)``
  returning an object with ``.ntotal`` — scripts/build_faiss_index.py:55-62,72
* ``search(query_emb, k) -> (distances[nq,k] f32 desc, indices[nq,k] i64, -1 padded)`` — src/serve/app.py:293-301
* ``save(dir)`` / ``load(dir)`` with ``index.faiss`` + ``doc_ids.json`` (+ ``texts.json`` read by the app) —
  scripts/build_faiss_index.py:66, src/serve/app.py:430-442, tests/conftest.py:188-198
* ``.doc_ids`` — src/serve/app.py:433

The reference builds an approximate HNSW graph; this backend answers with the
*exact* inner-product top-k (what the reference checks HNSW against: recall@10 >= 0.97,
configs/index.yaml:51-56), so ``index_type`` and the HNSW knobs are accepted and ignored.
All arithmetic runs in hand-written HIP kernels through the C-ABI; there is no CPU path.
"""
from __future__ import annotations

import json
import struct
from pathlib import Path
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _native

_FLAT_IP_FOURCC = b"IxFI"
_FAISS_DUMMY = 1 << 20
_METRIC_INNER_PRODUCT = 0


class IndexHandle:
    """What ``build_from_parquet`` returns: the reference only reads ``.ntotal`` (build_faiss_index.py:72)."""

    def __init__(self, owner: "FAISSIndexBuilder"):
        self._owner = owner

    @property
    def ntotal(self) -> int:
        return self._owner.ntotal

    @property
    def d(self) -> int:
        return self._owner.embedding_dim

    def search(self, x: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """faiss ``index.search`` shape: raw inner product, no query normalisation."""
        return self._owner._search_numpy(x, k, normalize_queries=False)


class FAISSIndexBuilder:
    """Exact cosine / inner-product index resident in MI355X HBM."""

    def __init__(
        self,
        embedding_dim: int = 384,
        index_type: str = "HNSW",
        metric: str = "cosine",
        device: Optional[str] = None,
        id_offset: int = 0,
    ) -> None:
        if embedding_dim != _native.SSKD_DIM:
            raise ValueError(
                f"embedding_dim={embedding_dim}: the gfx950 scan kernel is specialised for "
                f"{_native.SSKD_DIM}-d embeddings (e5-small-v2)"
            )
        metric = metric.lower()
        if metric not in ("cosine", "ip", "inner_product", "dot"):
            raise ValueError(f"metric={metric!r}: only cosine / inner product are supported")
        self.embedding_dim = embedding_dim
        self.index_type = index_type
        self.metric = "cosine" if metric == "cosine" else "ip"
        self.device = _resolve_device(device)
        self.doc_ids: List[str] = []
        self.doc_texts: Optional[dict] = None
        self.id_offset = int(id_offset)  # global id of local row 0 (row-sharded corpora)
        self.shard_info: Optional[dict] = None  # {"rank", "world_size", "n_total"} when written by build_sharded
        self._tiled: Optional[torch.Tensor] = None  # fp32 [capacity_rows * 384], tiled layout
        self._n = 0
        self._workspace: Optional[torch.Tensor] = None
        self.last_search_path: Optional[str] = None  # which path the last host search() took (diagnostic)
        # explicit launch tuning (``_native.SearchTuning``) handed to BOTH the workspace query and the
        # search call; ``None`` = the built-in plan.  There is no environment knob behind the search.
        self.search_tuning: Optional[_native.SearchTuning] = None
        # batch searches (k <= 10, >= 64 queries) go through the bf16-screened path: identical bits to
        # the exact scan (measured + proved error band, exact re-scoring, in-call exact fallback sized for
        # every query - no caller ever has to check a status), several times faster.
        # ``screening = False`` forces the plain exact scan.
        self.screening = True
        self._bf16: Optional[torch.Tensor] = None      # screening sidecar (bf16 tiles of the centred rows + norm block: 768 B per row), made lazily
        self._bf16_rows = -1
        # device int32[2] of the last screened search: [0] always 0, [1] = queries that took the in-call
        # exact fallback (a cost diagnostic; nothing to act on)
        self.last_status: Optional[torch.Tensor] = None
        self.index: Optional[IndexHandle] = None

    # ------------------------------------------------------------------ storage
    @property
    def ntotal(self) -> int:
        return self._n

    def _ensure_capacity(self, rows: int) -> None:
        lib = _native.load()
        need = int(lib.sskd_index_padded_rows(rows)) * self.embedding_dim
        have = 0 if self._tiled is None else self._tiled.numel()
        if need <= have:
            return
        _native.require_gpu()
        new_elems = max(need, int(have * 1.5))
        new = torch.empty(new_elems, dtype=torch.float32, device=self.device)
        if self._tiled is not None and self._n > 0:
            used = int(lib.sskd_index_padded_rows(self._n)) * self.embedding_dim
            new[:used].copy_(self._tiled[:used])
        self._tiled = new

    def reserve(self, rows: int) -> None:
        """Pre-size the HBM buffer (avoids regrowth while streaming a large corpus in)."""
        self._ensure_capacity(rows)

    def add(self, embeddings: Union[np.ndarray, torch.Tensor]) -> None:
        """Append rows (``faiss.normalize_L2`` + ``index.add``); host or device fp32 ``[n, 384]``.

        Rows are L2-normalised on the GPU when ``metric == "cosine"`` (configs/index.yaml:30).
        """
        lib = _native.load()
        x = _as_device_f32(embeddings, self.device)
        if x.dim() != 2 or x.shape[1] != self.embedding_dim:
            raise ValueError(f"expected [n, {self.embedding_dim}] embeddings, got {tuple(x.shape)}")
        n_new = x.shape[0]
        if n_new == 0:
            return
        tail = self._n % _native.SSKD_TILE_ROWS
        if tail:
            # rows cannot start mid-tile: re-pack the partial tail tile together with the new
            # rows as one contiguous [tail | new] block written at the tail tile's first row
            stream = _stream(self.device)
            staged = torch.empty((tail + n_new, self.embedding_dim), dtype=torch.float32, device=self.device)
            _native.check(
                lib.sskd_index_get_rows(self._tiled.data_ptr(), self._n - tail, tail, staged.data_ptr(), stream)
            )
            staged[tail:].copy_(x)
            if self.metric == "cosine":
                _native.check(
                    lib.sskd_l2_normalize_rows(staged[tail:].data_ptr(), n_new, self.embedding_dim, stream)
                )
            self._ensure_capacity(self._n + n_new)
            _native.check(
                lib.sskd_index_add_rows(
                    staged.data_ptr(), tail + n_new, 0, self._tiled.data_ptr(), self._n - tail, stream
                )
            )
        else:
            self._ensure_capacity(self._n + n_new)
            _native.check(
                lib.sskd_index_add_rows(
                    x.data_ptr(),
                    n_new,
                    1 if self.metric == "cosine" else 0,
                    self._tiled.data_ptr(),
                    self._n,
                    _stream(self.device),
                )
            )
        self._n += n_new
        self._bf16_rows = -1  # the bf16 screening copy is stale
        self.index = IndexHandle(self)

    def build_from_embeddings(
        self, embeddings: Union[np.ndarray, torch.Tensor], doc_ids: Optional[Sequence[str]] = None
    ) -> IndexHandle:
        self._n = 0
        self._tiled = None
        self.add(embeddings)
        self.doc_ids = list(doc_ids) if doc_ids is not None else [f"doc_{i}" for i in range(self._n)]
        if len(self.doc_ids) != self._n:
            raise ValueError(f"{len(self.doc_ids)} doc_ids for {self._n} vectors")
        self.index = IndexHandle(self)
        return self.index

    def build_from_parquet(
        self,
        model,
        parquet_path: Union[str, Path],
        batch_size: int = 32,
        max_docs: Optional[int] = None,
        hnsw_m: int = 32,
        hnsw_ef_construction: int = 200,
        text_column: str = "text",
        id_column: str = "chunk_id",
        show_progress: bool = True,
    ) -> IndexHandle:
        """Encode a parquet corpus with ``model.encode_documents`` and index it.

        Columns follow the reference's corpus schema (src/data/prepare.py:72-84,
        tests/conftest.py:210-216): ``text`` and ``chunk_id``.  ``hnsw_*`` are accepted for
        CLI compatibility (scripts/build_faiss_index.py:59-61); an exact scan has no graph.
        """
        del hnsw_m, hnsw_ef_construction
        ids, texts = read_corpus_parquet(parquet_path, max_docs, text_column, id_column)
        self._n = 0
        self._tiled = None
        self.reserve(len(texts))
        # stream in slabs so a multi-million-passage corpus never needs one host matrix
        slab = max(batch_size, 65536)
        on_device = getattr(model, "encode_documents_device", None)   # embeddings go encoder -> index tiles inside HBM
        for lo in range(0, len(texts), slab):
            if on_device is not None:
                embs = on_device(texts[lo : lo + slab], batch_size=batch_size)
            else:
                embs = model.encode_documents(texts[lo : lo + slab], batch_size=batch_size, show_progress=show_progress)
            self.add(embs)
        self.doc_ids = ids
        self.doc_texts = dict(zip(ids, texts))
        self.index = IndexHandle(self)
        return self.index

    # ------------------------------------------------------------------- search
    def search_device(
        self,
        queries: torch.Tensor,
        k: int,
        normalize_queries: Optional[bool] = None,
        out_scores: Optional[torch.Tensor] = None,
        out_ids: Optional[torch.Tensor] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Top-k over the index for device-resident queries; returns device tensors.

        No host synchronisation: everything is enqueued on the current stream.
        """
        lib = _native.load()
        if k < 1 or k > _native.SSKD_K_MAX:
            raise ValueError(f"k={k} outside [1, {_native.SSKD_K_MAX}]")
        if queries.dim() != 2 or queries.shape[1] != self.embedding_dim:
            raise ValueError(f"expected [nq, {self.embedding_dim}] queries, got {tuple(queries.shape)}")
        if queries.dtype != torch.float32 or not queries.is_cuda:
            raise TypeError("search_device expects a float32 device tensor")
        q = queries.contiguous()
        nq = q.shape[0]
        if normalize_queries is None:
            normalize_queries = self.metric == "cosine"
        stream = _stream(self.device)
        if normalize_queries and nq:
            q = q.clone()
            _native.check(lib.sskd_l2_normalize_rows(q.data_ptr(), nq, self.embedding_dim, stream))
        if out_scores is None:
            out_scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        if out_ids is None:
            out_ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        if nq == 0:
            return out_scores, out_ids
        tuning = self.search_tuning
        if self.screening and tuning is None and self._tiled is not None:
            need = int(lib.sskd_index_search_screened_workspace_bytes(self._n, nq, k))
            if need:
                if self._bf16 is None or self._bf16_rows != self._n:
                    self._bf16 = None
                    self._bf16 = torch.empty(int(lib.sskd_index_bf16_bytes(self._n)), dtype=torch.uint8, device=self.device)
                    _native.check(lib.sskd_index_make_bf16(self._tiled.data_ptr(), self._n, self._bf16.data_ptr(), stream))
                    self._bf16_rows = self._n
                if self._workspace is None or self._workspace.numel() < need:
                    self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
                self.last_status = torch.empty(2, dtype=torch.int32, device=self.device)
                _native.check(
                    lib.sskd_index_search_screened(
                        self._tiled.data_ptr(), self._bf16.data_ptr(), self._n, q.data_ptr(), nq, k, self.id_offset,
                        out_scores.data_ptr(), out_ids.data_ptr(), self.last_status.data_ptr(),
                        self._workspace.data_ptr(), self._workspace.numel(), stream, None, None,
                    )
                )
                return out_scores, out_ids
        self.last_status = None
        need = int(lib.sskd_index_search_workspace_bytes_ex(self._n, nq, k, tuning))
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        _native.check(
            lib.sskd_index_search_ex(
                0 if self._tiled is None else self._tiled.data_ptr(),
                self._n,
                q.data_ptr(),
                nq,
                k,
                self.id_offset,
                out_scores.data_ptr(),
                out_ids.data_ptr(),
                self._workspace.data_ptr(),
                self._workspace.numel(),
                stream,
                tuning,
                None,
                None,
            )
        )
        return out_scores, out_ids

    # the online shape: a handful of queries (sskd_amd.h, one-pass variant).  With a single query
    # block the shared pruning pools of the batch kernel only cost, so the pool-free one-pass
    # search is also the faster one for small k (where its proof cannot fail for k <= 10).
    ONEPASS_MAX_NQ = 64
    ONEPASS_MAX_K = 256

    def _search_onepass_device(self, q: torch.Tensor, k: int, normalize_queries: bool):
        """One corpus pass + proof of exactness; returns ``(scores, ids, inexact_flag)`` device tensors."""
        lib = _native.load()
        stream = _stream(self.device)
        nq = q.shape[0]
        if normalize_queries:
            q = q.clone()
            _native.check(lib.sskd_l2_normalize_rows(q.data_ptr(), nq, self.embedding_dim, stream))
        out_scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        flag = torch.empty(1, dtype=torch.int32, device=self.device)
        need = int(lib.sskd_index_search_onepass_workspace_bytes(self._n, nq, k))
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        _native.check(
            lib.sskd_index_search_onepass(
                self._tiled.data_ptr(), self._n, q.data_ptr(), nq, k, self.id_offset,
                out_scores.data_ptr(), out_ids.data_ptr(), flag.data_ptr(),
                self._workspace.data_ptr(), self._workspace.numel(), stream,
            )
        )
        return out_scores, out_ids, flag

    def _search_numpy(self, query_emb, k: int, normalize_queries: Optional[bool]) -> Tuple[np.ndarray, np.ndarray]:
        _native.require_gpu()
        q = np.ascontiguousarray(np.asarray(query_emb, dtype=np.float32))
        if q.ndim == 1:
            q = q[None, :]
        qd = torch.from_numpy(q).to(self.device)
        with torch.cuda.device(self.device):
            nq = qd.shape[0]
            if (
                1 <= k <= self.ONEPASS_MAX_K
                and 1 <= nq <= self.ONEPASS_MAX_NQ
                and self._n >= 1
                and qd.shape[1] == self.embedding_dim
            ):
                norm = self.metric == "cosine" if normalize_queries is None else normalize_queries
                scores, ids, flag = self._search_onepass_device(qd, k, norm)
                # this host path synchronises anyway (NumPy out): read the proof flag with the result
                if int(flag.item()) == 0:
                    self.last_search_path = "onepass"
                    return scores.cpu().numpy(), ids.cpu().numpy()
                self.last_search_path = "onepass-unproven+chained"
            else:
                self.last_search_path = "chained" if k > _native.SSKD_K_PASS else "single"
            scores, ids = self.search_device(qd, k, normalize_queries=normalize_queries)
            if self.last_status is not None:
                self.last_search_path += "+screened"
            return scores.cpu().numpy(), ids.cpu().numpy()

    def search(self, query_emb: np.ndarray, k: int = 10) -> Tuple[np.ndarray, np.ndarray]:
        """``(distances, indices)`` exactly as the serving route consumes them (app.py:293-301)."""
        if self._n == 0 and self._tiled is None and self.index is None:
            raise RuntimeError("index is empty: call build_from_parquet/add/load first")
        return self._search_numpy(query_emb, k, normalize_queries=None)

    # -------------------------------------------------------------- persistence
    def reconstruct(self, rows: Sequence[int]) -> np.ndarray:
        """Stored vectors of the given local rows (``faiss.Index.reconstruct`` for a list), host fp32."""
        lib = _native.load()
        rows = [int(r) for r in rows]
        out = torch.empty((len(rows), self.embedding_dim), dtype=torch.float32, device=self.device)
        for j, r in enumerate(rows):
            if not 0 <= r < self._n:
                raise IndexError(f"row {r} outside [0, {self._n})")
            _native.check(lib.sskd_index_get_rows(self._tiled.data_ptr(), r, 1, out[j].data_ptr(), _stream(self.device)))
        return out.cpu().numpy()

    def to_numpy(self) -> np.ndarray:
        """Row-major copy of the stored vectors (host)."""
        lib = _native.load()
        out = torch.empty((self._n, self.embedding_dim), dtype=torch.float32, device=self.device)
        if self._n:
            _native.check(
                lib.sskd_index_get_rows(self._tiled.data_ptr(), 0, self._n, out.data_ptr(), _stream(self.device))
            )
        return out.cpu().numpy()

    def save(self, output_dir: Union[str, Path]) -> None:
        """Write ``index.faiss`` (flat inner-product layout) + ``doc_ids.json`` (+ ``texts.json``).  A row shard
        (``id_offset != 0``, or one written by ``sharded_index.build_sharded``) also gets ``shard.json`` with the
        global id of its first row, which ``load`` restores."""
        out = Path(output_dir)
        out.mkdir(parents=True, exist_ok=True)
        write_flat_ip(out / "index.faiss", self.to_numpy())
        with open(out / "doc_ids.json", "w") as f:
            json.dump(list(self.doc_ids), f)
        if self.doc_texts is not None:
            with open(out / "texts.json", "w") as f:
                json.dump(self.doc_texts, f)
        if self.id_offset != 0 or self.shard_info:
            with open(out / "shard.json", "w") as f:
                json.dump({**(self.shard_info or {}), "id_offset": self.id_offset, "rows": self._n}, f)

    def load(self, index_dir: Union[str, Path], append: bool = False) -> None:
        """Restore a saved index (``append=True``: add its rows behind the ones already held - consecutive row
        shards served by one process).  ``shard.json``, when present, restores ``id_offset``."""
        d = Path(index_dir)
        path = d / "index.faiss"
        if not path.exists():
            raise FileNotFoundError(f"{path} not found")
        vecs = read_flat_ip(path)
        if vecs.shape[1] != self.embedding_dim:
            raise ValueError(f"index has dim {vecs.shape[1]}, builder expects {self.embedding_dim}")
        shard_path = d / "shard.json"
        shard = json.loads(shard_path.read_text()) if shard_path.exists() else None
        metric, self.metric = self.metric, "ip"  # stored vectors are already normalised
        try:
            if not append:
                self._n = 0
                self._tiled = None
                self.doc_ids = []
                self.doc_texts = None
                if shard is not None:
                    self.id_offset = int(shard["id_offset"])
                    self.shard_info = {k: v for k, v in shard.items() if k not in ("id_offset", "rows")}
            elif shard is not None and int(shard["id_offset"]) != self.id_offset + self._n:
                raise ValueError(f"{d}: shard starts at row {shard['id_offset']}, expected {self.id_offset + self._n}")
            first = self._n
            self.reserve(self._n + vecs.shape[0])
            step = 1 << 18  # stream the (memory-mapped) matrix in 400 MB slabs
            for lo in range(0, vecs.shape[0], step):
                self.add(np.array(vecs[lo : lo + step], dtype=np.float32, copy=True))
        finally:
            self.metric = metric
        ids_path = d / "doc_ids.json"
        if ids_path.exists():
            with open(ids_path) as f:
                self.doc_ids = list(self.doc_ids) + list(json.load(f))
        else:
            self.doc_ids = list(self.doc_ids) + [f"doc_{self.id_offset + i}" for i in range(first, self._n)]
        texts_path = d / "texts.json"
        if texts_path.exists():
            with open(texts_path) as f:
                self.doc_texts = {**(self.doc_texts or {}), **json.load(f)}
        self.index = IndexHandle(self)

    def cleanup(self) -> None:
        self._tiled = None
        self._workspace = None
        self._n = 0


# ---------------------------------------------------------------------- helpers
def read_corpus_parquet(parquet_path, max_docs: Optional[int] = None, text_column: str = "text",
                        id_column: str = "chunk_id") -> Tuple[List[str], List[str]]:
    """``(ids, texts)`` of a corpus in the reference's schema (src/data/prepare.py:72-84, tests/conftest.py:210-216)."""
    import pandas as pd

    df = pd.read_parquet(parquet_path)
    if max_docs is not None:
        df = df.head(max_docs)
    if text_column not in df.columns:
        raise KeyError(f"parquet file {parquet_path} has no {text_column!r} column")
    texts = df[text_column].astype(str).tolist()
    if id_column in df.columns:
        ids = df[id_column].astype(str).tolist()
    else:
        ids = [f"doc_{i}" for i in range(len(texts))]
    return ids, texts


def _resolve_device(device: Optional[str]) -> torch.device:
    if device is None or device == "cuda":
        return torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"device={device!r}: semantic-search-kd_amd runs on MI355X only (PyTorch-ROCm spells it 'cuda[:N]'); "
            "there is no CPU path"
        )
    return dev


def _stream(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def _as_device_f32(x: Union[np.ndarray, torch.Tensor], device: torch.device) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    _native.require_gpu()
    return t.to(device=device, dtype=torch.float32).contiguous()


def write_flat_ip(path: Union[str, Path], vectors: np.ndarray) -> None:
    """Serialise vectors in faiss' ``IndexFlatIP`` layout (fourcc ``IxFI``, faiss/impl/index_write.cpp)."""
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    n, d = v.shape
    with open(path, "wb") as f:
        f.write(_FLAT_IP_FOURCC)
        f.write(struct.pack("<iqqqBi", d, n, _FAISS_DUMMY, _FAISS_DUMMY, 1, _METRIC_INNER_PRODUCT))
        f.write(struct.pack("<Q", n * d))
        v.tofile(f)


def read_flat_ip(path: Union[str, Path]) -> np.ndarray:
    """Memory-map the vectors of a flat inner-product index file written by :func:`write_flat_ip`."""
    with open(path, "rb") as f:
        fourcc = f.read(4)
        if fourcc != _FLAT_IP_FOURCC:
            raise ValueError(
                f"{path}: index type {fourcc!r} is not a flat inner-product index; HNSW graph files "
                "cannot be loaded by the exact-scan backend — rebuild the index from the corpus"
            )
        d, n, _, _, _, metric = struct.unpack("<iqqqBi", f.read(4 + 8 * 3 + 1 + 4))
        (count,) = struct.unpack("<Q", f.read(8))
        offset = f.tell()
    if metric != _METRIC_INNER_PRODUCT or count != n * d:
        raise ValueError(f"{path}: malformed flat index header (d={d}, n={n}, count={count}, metric={metric})")
    if n == 0:
        return np.zeros((0, d), dtype=np.float32)
    return np.memmap(path, dtype=np.float32, mode="r", offset=offset, shape=(n, d))
