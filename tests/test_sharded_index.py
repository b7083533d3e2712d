"""Product-level sharding: build -> persist -> reload -> serve a row-sharded index (sharded_index.py).

CPU (``gloo``, world 2 / 4): the whole protocol - every rank encodes and saves its own shard, rank 0 writes the
manifest, a DIFFERENT world size reloads it (8 shards -> 1, 2 or 4 ranks), rank 0 searches while the other ranks sit
in ``serve_forever``, ``/index/load``-style reload, shutdown - with the HBM index replaced by an oracle-backed
stand-in (tests may use the oracle; there is no CPU product path).  GPU (``-m gpu``): the same through the real
``FAISSIndexBuilder`` shards and the build CLI under ``torch.distributed.run``, two ranks.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search as oracle
from semantic_search_kd_amd.dist import shard_bounds
from semantic_search_kd_amd.index import read_flat_ip, write_flat_ip
from semantic_search_kd_amd.sharded_index import MANIFEST, ShardedIndex, build_sharded, open_index

N_DOCS = 203


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _corpus(tmp: Path) -> Path:
    path = tmp / "corpus.parquet"
    if not path.exists():
        pd.DataFrame({"chunk_id": [f"chunk_{i}" for i in range(N_DOCS)], "text": [f"t{i}" for i in range(N_DOCS)]}).to_parquet(path)
    return path


_ROWS = oracle.seeded_unit_rows(N_DOCS, 384, 99)


class SeededModel:
    """duck-typed StudentModel: text "t<i>" -> seeded unit row i (the encoder is not what is under test here)"""

    embedding_dim = 384

    def encode_documents(self, docs, batch_size=32, show_progress=False):
        return _ROWS[[int(d[1:]) for d in docs]]


class OracleIndex:
    """Stand-in for FAISSIndexBuilder on CPU: same surface as far as sharded_index.py uses it, oracle arithmetic."""

    def __init__(self, embedding_dim, metric, device, id_offset):
        self.embedding_dim, self.metric, self.device, self.id_offset = embedding_dim, metric, "cpu", id_offset
        self.rows = np.zeros((0, embedding_dim), np.float32)
        self.doc_ids, self.doc_texts, self.shard_info = [], None, None

    def reserve(self, n):
        pass

    def add(self, x):
        self.rows = np.concatenate([self.rows, np.asarray(x, np.float32)])

    def save(self, out):
        out = Path(out)
        out.mkdir(parents=True, exist_ok=True)
        write_flat_ip(out / "index.faiss", self.rows)
        (out / "doc_ids.json").write_text(json.dumps(self.doc_ids))
        (out / "texts.json").write_text(json.dumps(self.doc_texts))
        (out / "shard.json").write_text(json.dumps({**self.shard_info, "id_offset": self.id_offset, "rows": len(self.rows)}))

    def load(self, d, append=False):
        d = Path(d)
        shard = json.loads((d / "shard.json").read_text())
        if not append:
            self.rows = np.zeros((0, self.embedding_dim), np.float32)
            self.id_offset = shard["id_offset"]
        else:
            assert shard["id_offset"] == self.id_offset + len(self.rows)
        self.add(np.array(read_flat_ip(d / "index.faiss")))

    def search_device(self, q, k, normalize_queries=None, out_scores=None, out_ids=None):
        s, i = oracle.topk_fma(q.numpy(), self.rows, k, id_offset=self.id_offset)
        return torch.from_numpy(s), torch.from_numpy(i)

    def cleanup(self):
        pass


def _oracle_merge(all_s, all_i, k):
    s, i = oracle.topk_merge(all_s.numpy(), all_i.numpy(), k)
    return torch.from_numpy(s), torch.from_numpy(i)


def _factory(embedding_dim, metric, device, id_offset):
    return OracleIndex(embedding_dim, metric, device, id_offset)


def _queries():
    return oracle.seeded_unit_rows(9, 384, 5)


def _build_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        manifest = build_sharded(SeededModel(), _corpus(Path(tmp)), Path(tmp) / f"index_w{world}", batch_size=16,
                                 index_factory=_factory)
        assert manifest["n_total"] == N_DOCS and len(manifest["shards"]) == world
    finally:
        dist.destroy_process_group()


def _serve_worker(rank, world, port, tmp, index_dir, second_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        index = ShardedIndex(index_factory=_factory)
        index.load_all_ranks(index_dir)
        index._searcher.merge = _oracle_merge
        if rank != 0:
            orig = index._load_local

            def reload(d):   # the reload announced by rank 0 builds a new searcher: give it the CPU merge again
                orig(d)
                index._searcher.merge = _oracle_merge

            index._load_local = reload
            index.serve_forever()
            return
        ref_s, ref_i = oracle.topk_fma(_queries(), _ROWS, 10)
        s, i = index.search(_queries(), 10)
        assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
        s1, i1 = index.search(_queries()[0], 3)          # one query, 1-D, as /search hands it over
        assert np.array_equal(i1, ref_i[:1, :3])
        assert index.doc_ids == [f"chunk_{j}" for j in range(N_DOCS)] and index.doc_texts["chunk_7"] == "t7"
        assert index.ntotal == N_DOCS
        # hot reload (the /index/load route): the waiting ranks follow rank 0
        index.load(second_dir)
        index._searcher.merge = _oracle_merge
        s, i = index.search(_queries(), 10)
        assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
        index.close()
        np.save(Path(tmp) / f"served_w{world}.npy", i)
    finally:
        dist.destroy_process_group()


def test_build_save_reload_serve_sharded_gloo(tmp_path):
    tmp = str(tmp_path)
    _corpus(tmp_path)
    # build with 4 ranks and, separately, with 2
    for world in (4, 2):
        mp.spawn(_build_worker, args=(world, _free_port(), tmp), nprocs=world, join=True)
    d4, d2 = tmp_path / "index_w4", tmp_path / "index_w2"
    man = json.loads((d4 / MANIFEST).read_text())
    assert [s["id_offset"] for s in man["shards"]] == [shard_bounds(N_DOCS, 4, r)[0] for r in range(4)]
    assert sum(s["rows"] for s in man["shards"]) == N_DOCS
    # every shard directory is by itself a loadable index whose shard.json carries its id offset
    sh = json.loads((d4 / "shard_2" / "shard.json").read_text())
    assert sh["id_offset"] == shard_bounds(N_DOCS, 4, 2)[0] and sh["world_size"] == 4 and sh["n_total"] == N_DOCS
    assert np.array_equal(np.array(read_flat_ip(d4 / "shard_2" / "index.faiss")), _ROWS[slice(*shard_bounds(N_DOCS, 4, 2))])
    # serve the 4-shard index with 2 ranks (two shards each), then hot-reload the 2-shard one
    mp.spawn(_serve_worker, args=(2, _free_port(), tmp, str(d4), str(d2)), nprocs=2, join=True)
    assert (tmp_path / "served_w2.npy").exists()
    # ... with 4 ranks, and with ONE process holding all four shards (no process group at all)
    mp.spawn(_serve_worker, args=(4, _free_port(), tmp, str(d4), str(d2)), nprocs=4, join=True)
    single = ShardedIndex(index_factory=_factory)
    single.load(d4)
    s, i = single.search(_queries(), 10)
    ref_s, ref_i = oracle.topk_fma(_queries(), _ROWS, 10)
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s) and single.local.id_offset == 0
    assert len(single.local.rows) == N_DOCS


def test_more_ranks_than_shards_and_open_index_dispatch(tmp_path):
    _corpus(tmp_path)
    mp.spawn(_build_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    d2 = tmp_path / "index_w2"
    # 4 ranks over 2 shards: ranks 0 and 2 get... contiguous runs [r S / G, (r+1) S / G): two ranks stay empty
    mp.spawn(_serve_worker, args=(4, _free_port(), str(tmp_path), str(d2), str(d2)), nprocs=4, join=True)
    from semantic_search_kd_amd.sharded_index import is_sharded_dir

    assert is_sharded_dir(d2) and not is_sharded_dir(d2 / "shard_0")
    cur = ShardedIndex(index_factory=_factory)
    assert open_index(d2, 384, current=cur) is cur and cur.ntotal == N_DOCS


# ----------------------------------------------------------------------------- GPU: the real shards
def _gpu_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    multi = torch.cuda.device_count() >= world
    dev = f"cuda:{rank if multi else 0}"
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if multi else "gloo", rank=rank, world_size=world,
                            **({"device_id": torch.device(dev)} if multi else {}))
    try:
        out = Path(tmp) / "gpu_index"
        # metric "ip": rows and queries are used as given (no re-normalisation), so the oracle sees the same bits
        build_sharded(SeededModel(), _corpus(Path(tmp)), out, batch_size=16, device=dev, metric="ip")
        index = ShardedIndex(device=dev)
        index.load_all_ranks(out)
        assert index.local.id_offset == shard_bounds(N_DOCS, world, rank)[0]
        assert index.local.ntotal == shard_bounds(N_DOCS, world, rank)[1] - shard_bounds(N_DOCS, world, rank)[0]
        if rank != 0:
            index.serve_forever()
            return
        s, i = index.search(_queries(), 10)
        ref_s, ref_i = oracle.topk_fma(_queries(), _ROWS, 10)
        assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
        big = oracle.seeded_unit_rows(70, 384, 6)          # >= 64 queries: the screened path inside each shard... needs
        s, i = index.search(big, 10)                        # >= 2048 rows per shard to engage; here the exact scan serves
        ref_s, ref_i = oracle.topk_fma(big, _ROWS, 10)
        assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
        index.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_build_reload_serve_two_ranks_gpu(gpu, tmp_path):
    """build_sharded -> shard_r/ on disk -> ShardedIndex.load -> rank 0 searches, rank 1 serves: real HBM shards,
    bit-identical to the oracle over the whole corpus; then ONE process reloads both shards into one buffer."""
    _corpus(tmp_path)
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    single = ShardedIndex(device="cuda:0")
    single.load(tmp_path / "gpu_index")
    assert single.local.ntotal == N_DOCS and single.local.id_offset == 0
    s, i = single.search(_queries(), 10)
    ref_s, ref_i = oracle.topk_fma(_queries(), _ROWS, 10)
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)
    # a shard directory alone is a FAISSIndexBuilder.load target that keeps its global ids
    from semantic_search_kd_amd import FAISSIndexBuilder

    b = FAISSIndexBuilder(embedding_dim=384, metric="ip")
    b.load(tmp_path / "gpu_index" / "shard_1")
    lo, hi = shard_bounds(N_DOCS, 2, 1)
    assert b.id_offset == lo and b.ntotal == hi - lo
    s, i = b.search(_queries(), 5)
    ref_s, ref_i = oracle.topk_fma(_queries(), _ROWS[lo:hi], 5, id_offset=lo)
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)


@pytest.mark.gpu
def test_build_index_cli_under_torchrun_two_ranks(gpu, tmp_path):
    """The reference's build CLI (scripts/build_faiss_index.py:14-73) launched under torch.distributed.run with two
    ranks: each writes its shard, rank 0 the manifest; the served result equals the single-process index's."""
    from semantic_search_kd_amd import BertConfig, FAISSIndexBuilder, StudentModel, synthetic_state_dict
    from semantic_search_kd_amd.weights import save_model_dir
    from test_encoder_gpu import _vocab

    vocab = _vocab()
    cfg = BertConfig(vocab_size=len(vocab), num_hidden_layers=2)
    mdir = tmp_path / "model"
    save_model_dir(mdir, cfg, synthetic_state_dict(cfg))
    (mdir / "vocab.txt").write_text("\n".join(vocab))
    docs = [" ".join(vocab[5 + (i * 7 + j) % (len(vocab) - 5)] for j in range(3 + i % 9)) for i in range(41)]
    corpus = tmp_path / "c.parquet"
    pd.DataFrame({"chunk_id": [f"c{i}" for i in range(len(docs))], "text": docs}).to_parquet(corpus)
    out = tmp_path / "idx"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=str(Path(__file__).resolve().parent.parent))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "semantic_search_kd_amd.build_index_cli", "--model-path", str(mdir),
           "--data-path", str(corpus), "--output-dir", str(out), "--batch-size", "8", "--device", "cuda:0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Total vectors: 41 in 2 shard(s)" in r.stdout
    student = StudentModel(str(mdir), device="cuda:0")
    whole = FAISSIndexBuilder(embedding_dim=384, metric="cosine", device="cuda:0")
    whole.build_from_parquet(model=student, parquet_path=corpus, batch_size=8)
    sharded = open_index(out, 384, device="cuda:0")
    assert isinstance(sharded, ShardedIndex) and sharded.doc_ids == whole.doc_ids and sharded.ntotal == 41
    q = student.encode_queries([docs[3], docs[17], "zzz"])
    s1, i1 = sharded.search(q, 5)
    s2, i2 = whole.search(q, 5)
    # the two builds encode a passage in different launches (different batch-mates): scores agree to bf16-encoder
    # tolerance, and ids wherever the ranking is not a near-tie
    assert np.abs(s1 - s2).max() <= 4e-3
    for row in range(3):   # a random-init encoder puts several passages within 1e-3 of each other: compare as sets
        assert i1[row, 0] in i2[row] and i2[row, 0] in i1[row], (i1[row], i2[row])
    assert 3 in i1[0] and 17 in i1[1]
