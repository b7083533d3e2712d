#!/usr/bin/env python3
"""Headline benchmark: queries/s @ top-10 over a 1M x 384-d corpus (+ docs embedded/s).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
For N > 1 the driver launches it under ``python -m torch.distributed.run`` (one rank per GPU over
RCCL); started from a bare shell without ``RANK`` in the environment it starts that launcher itself
as a CHILD process - before anything touches the GPU - and exits with the child's return code.

Workload = BASELINE.json configs[1]:
  search : corpus 1 000 000 x 384 fp32 unit rows (seed 1234) resident in HBM, row-sharded over
           the N ranks; 10 000 queries (seed 4321) replicated; k = 10.  One *step* = all 10 000
           queries answered: local exact scan -> (N > 1) all-gather of partial top-10 -> merge.
           Fixed total work as N grows => "strong" scaling.
  encode : e5-small-v2-shaped bf16 encoder, batch 512 x seq 256 synthetic token ids per rank
           (timed separately in the same run once the encoder kernels are built).
``value`` is queries/s of the search step (whole job); docs/s rides along in ``encode``.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

METRIC = "docs embedded/sec + queries/sec@top-10 (1M×384-d corpus), 1→8 MI355X"
N_CORPUS = 1_000_000
N_QUERIES = 10_000
K = 10
DIM = 384
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA peak (= vector rate)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


class HipEvents:
    """hipEvent pairs recorded by the C-ABI around the scan kernel on the launch stream."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]

    def create(self) -> C.c_void_p:
        ev = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(ev)) == 0
        return ev

    def elapsed_ms(self, a, b) -> float:
        ms = C.c_float()
        assert self.hip.hipEventSynchronize(b) == 0
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return float(ms.value)


def physical_cores() -> int:
    """Physical cores this process may use (the CPU baselines run one BLAS / OpenMP thread per core)."""
    try:
        import psutil

        phys = psutil.cpu_count(logical=False) or 0
    except Exception:
        phys = 0
    try:
        allowed = len(os.sched_getaffinity(0))
    except Exception:
        allowed = os.cpu_count() or 1
    phys = phys or allowed
    return max(1, min(phys, allowed))


def cpu_search_baseline(n_rows: int, n_queries: int, k: int, budget_s: float = 20.0):
    """The reference's CPU exact-search idiom on the box's host cores (kind "port"): ONE ``np.matmul`` over a whole
    block of queries, as scripts/simple_eval.py:25 multiplies all its queries at once, then top-k
    (argpartition + sort of the k survivors instead of the reference's full argsort of 10^6 scores per query,
    scripts/simple_eval.py:35 - the baseline is not made slower than it has to be).  Synthetic unit rows of
    the GPU leg's shape, drawn on the host.  Bounded sample: as many 1 024-query blocks as fit ~budget_s."""
    from oracle import search as oracle  # checker / baseline leg only

    rng = np.random.default_rng(1234)
    corpus = rng.standard_normal((n_rows, DIM), dtype=np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    queries = rng.standard_normal((min(n_queries, 8192), DIM), dtype=np.float32)
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    from concurrent.futures import ThreadPoolExecutor

    oracle.scores_blas(queries[:64], corpus[:4096])   # BLAS thread pool up before the clock starts
    pool = ThreadPoolExecutor(max_workers=physical_cores())   # the top-k of a score block, rows split over the cores
    done, t0 = 0, time.perf_counter()
    batch, chunk = 1024, 131072
    while done < queries.shape[0]:
        q = queries[done : done + batch]
        best_s = np.full((q.shape[0], k), -np.inf, np.float32)
        best_i = np.full((q.shape[0], k), -1, np.int64)
        for lo in range(0, n_rows, chunk):
            s = oracle.scores_blas(q, corpus[lo : lo + chunk])
            bounds = np.linspace(0, s.shape[0], min(physical_cores(), s.shape[0]) + 1).astype(int)
            parts = list(pool.map(lambda ab: np.argpartition(-s[ab[0] : ab[1]], k - 1, axis=1)[:, :k],
                                  zip(bounds[:-1], bounds[1:])))   # numpy's partition releases the GIL
            part = np.concatenate(parts, axis=0)
            cand_s = np.concatenate([best_s, np.take_along_axis(s, part, axis=1)], axis=1)
            cand_i = np.concatenate([best_i, part + lo], axis=1)
            order = np.argsort(-cand_s, axis=1, kind="stable")[:, :k]
            best_s = np.take_along_axis(cand_s, order, axis=1)
            best_i = np.take_along_axis(cand_i, order, axis=1)
        done += q.shape[0]
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, dt


def cpu_encode_baseline(batch: int = 32, seq_len: int = 256, max_docs: int = 1000, budget_s: float = 15.0):
    """The reference's encode path on the box's host cores: ``transformers.BertModel`` (what
    sentence-transformers executes for StudentModel.encode) + mean-pool + L2-normalise, fp32, batch 32
    (the reference CLI default, scripts/build_faiss_index.py:20), same synthetic weights and token
    shape as the GPU leg; bounded sample (~budget_s).  Without ``transformers`` on the box the
    repo's torch-CPU restatement (oracle/encoder.py) is timed instead (kind "port")."""
    import importlib.util

    from oracle import encoder as enc_oracle  # checker / baseline leg only
    from semantic_search_kd_amd.weights import BertConfig, synthetic_state_dict

    cfg = BertConfig()
    sd = synthetic_state_dict(cfg)
    ids, mask = enc_oracle.synthetic_token_ids(batch, seq_len, seed=0)
    engine = "oracle/encoder.py torch-CPU restatement"
    model = None
    if importlib.util.find_spec("transformers") is not None:
        try:
            from transformers import BertConfig as HFConfig
            from transformers import BertModel

            hf_cfg = HFConfig(
                vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
                layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu", hidden_dropout_prob=0.0,
                attention_probs_dropout_prob=0.0,
            )
            model = BertModel(hf_cfg, add_pooling_layer=False).eval()
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
            engine = "transformers.BertModel"
        except Exception:
            model = None
    tid, tmask = torch.from_numpy(ids).long(), torch.from_numpy(mask).long()

    def one_batch():
        if model is not None:
            with torch.no_grad():
                h = model(input_ids=tid, attention_mask=tmask).last_hidden_state.numpy()
            return enc_oracle.mean_pool_normalize(h, mask, True)
        return enc_oracle.encode_token_ids(sd, ids, mask, cfg.num_hidden_layers)

    one_batch()  # warm-up (thread pools, allocator)
    # thread count: one per physical core is not the fastest for a 33 M-parameter model at batch 32 on a many-core host
    # (synchronisation dominates): take the best of a few counts, so the baseline is not handicapped by the setting
    cores = torch.get_num_threads()
    best_threads, best_t = cores, None
    for nt in sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 16), min(cores, 8)}):
        torch.set_num_threads(nt)
        one_batch()
        t1 = time.perf_counter()
        one_batch()
        dt1 = time.perf_counter() - t1
        if best_t is None or dt1 < best_t:
            best_threads, best_t = nt, dt1
    torch.set_num_threads(best_threads)
    done, t0 = 0, time.perf_counter()
    while done < max_docs:
        e = one_batch()
        done += batch
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    assert np.isfinite(e).all()
    from semantic_search_kd_amd.bench_support import encoder_flops

    return {
        "value": round(done / dt, 2),
        "unit": "docs/s",
        "gflops": round(encoder_flops(done * seq_len, seq_len, cfg) / dt / 1e9, 1),
        "cores": torch.get_num_threads(),
        "kind": "reference" if model is not None else "port",
        "sample": f"{done} passages of {seq_len} tokens in batches of {batch} ({engine}, fp32, the GPU leg's "
                  f"synthetic weights), {dt:.1f} s",
        "host_cpus": os.cpu_count(),
    }


def faiss_probe() -> str:
    """the reference's HNSW path is timed only when faiss is importable on the box."""
    import importlib.util

    return "available" if importlib.util.find_spec("faiss") is not None else \
        "faiss not installed on this box: the reference's IndexHNSWFlat path cannot be timed; " \
        "cpu_baseline is the exact IndexFlatIP idiom (numpy sgemm + top-k)"


def cpu_baselines_child() -> None:
    """``bench.py --cpu-baseline-child``: both CPU baselines in a FRESH process that never touches the GPU, one
    thread per physical core (set before NumPy / torch start their pools), printed as one JSON line."""
    cores = physical_cores()
    torch.set_num_threads(cores)
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-baseline-child", action="store_true")
    ap.add_argument("--corpus", type=int, default=N_CORPUS)
    ap.add_argument("--queries", type=int, default=N_QUERIES)
    ap.add_argument("--no-encode", action="store_true")
    args, _ = ap.parse_known_args()
    out = {}
    qps, cpu_nq, cpu_dt = cpu_search_baseline(args.corpus, args.queries, K)
    threads = cores
    try:
        from threadpoolctl import threadpool_info

        blas = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
        threads = max(blas) if blas else threads
    except Exception:
        pass
    out["search"] = {
        "value": round(qps, 2),
        "unit": "queries/s",
        "gflops": round(2.0 * qps * args.corpus * DIM / 1e9, 1),
        "cores": threads,
        "kind": "port",
        "sample": f"{cpu_nq} queries in blocks of 1024 against a {args.corpus}-row synthetic corpus (one numpy sgemm per "
                  f"block and 131072-row slab + argpartition: the reference's exact-search idiom, scripts/simple_eval.py:25,35), "
                  f"{cpu_dt:.1f} s, fresh process before any GPU work",
        "host_cpus": os.cpu_count(),
        "physical_cores": cores,
        "faiss_hnsw": faiss_probe(),
    }
    if not args.no_encode:
        out["encode"] = cpu_encode_baseline()
        out["encode"]["physical_cores"] = cores
    print("CPU_BASELINES " + json.dumps(out), flush=True)


def run_cpu_baselines_first(args) -> dict:
    """Start the child BEFORE this process initialises the GPU (so no GPU-leg thread pool competes with it and none
    of its own pools linger into the GPU legs) and wait for it: 30-40 s."""
    import subprocess

    cores = physical_cores()
    env = dict(os.environ)
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        env[var] = str(cores)
    cmd = [sys.executable, str(Path(__file__).resolve()), "--cpu-baseline-child", "--corpus", str(args.corpus),
           "--queries", str(args.queries)] + (["--no-encode"] if args.no_encode else [])
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        for ln in r.stdout.splitlines():
            if ln.startswith("CPU_BASELINES "):
                return json.loads(ln[len("CPU_BASELINES "):])
        return {"error": (r.stderr or r.stdout)[-400:]}
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


def search_hip_sha() -> str:
    import hashlib

    return hashlib.sha256((REPO / "semantic-search-kd_amd" / "csrc" / "search.hip").read_bytes()).hexdigest()[:16]


def _time_search(index, queries, k, steps, ev, lib, n_local, id_offset, screened: bool):
    """(ms per call, kernel ms from HIP events around the dominant kernel, status) of the device-resident search"""
    from semantic_search_kd_amd import _native

    dev = queries.device
    nq = queries.shape[0]
    out_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    st_ptr = int(torch.cuda.current_stream(dev).cuda_stream)
    if screened:
        index.search_device(queries[:64], k, normalize_queries=False)   # builds the screening sidecar
        ws = torch.empty(int(lib.sskd_index_search_screened_workspace_bytes(n_local, nq, k)), dtype=torch.uint8, device=dev)
    else:
        ws = torch.empty(int(lib.sskd_index_search_workspace_bytes(n_local, nq, k)), dtype=torch.uint8, device=dev)
    pairs = [(ev.create(), ev.create()) for _ in range(steps)]

    def call(a, b):
        if screened:
            _native.check(lib.sskd_index_search_screened(index._tiled.data_ptr(), index._bf16.data_ptr(), n_local, queries.data_ptr(),
                                                         nq, k, id_offset, out_s.data_ptr(), out_i.data_ptr(), status.data_ptr(),
                                                         ws.data_ptr(), ws.numel(), st_ptr, a, b))
        else:
            _native.check(lib.sskd_index_search_profiled(index._tiled.data_ptr(), n_local, queries.data_ptr(), nq, k, id_offset,
                                                         out_s.data_ptr(), out_i.data_ptr(), ws.data_ptr(), ws.numel(), st_ptr, a, b))

    call(None, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in pairs:
        call(a, b)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    kernel_ms = float(np.mean([ev.elapsed_ms(a, b) for a, b in pairs]))
    return ms, kernel_ms, status.cpu().numpy(), out_s, out_i


def bench_search_anisotropic(pkg, lib, dev, n, nq, ev, steps: int = 5):
    """The search on HOSTILE data instead of isotropic random unit vectors: e5-like geometry (a common component
    that puts the mean pairwise cosine at 0.8), 64 topical clusters, 1 % near-duplicate rows, and a fifth of the
    queries planted next to corpus rows.  Reports queries/s, how many queries needed the in-call exact fallback,
    and that every output row equals the exact scan's (all queries, bit for bit)."""
    g = torch.Generator(device=dev).manual_seed(99)
    common = torch.nn.functional.normalize(torch.randn(DIM, generator=g, device=dev), dim=0)
    centres = torch.randn((64, DIM), generator=g, device=dev) / DIM ** 0.5
    index = pkg.FAISSIndexBuilder(embedding_dim=DIM, index_type="HNSW", metric="ip", device=str(dev))
    index.reserve(n)
    first = None
    for lo in range(0, n, 1 << 18):
        m = min(1 << 18, n - lo)
        rows = 2.0 * common + 0.5 * centres[torch.randint(0, 64, (m,), generator=g, device=dev)] \
            + torch.randn((m, DIM), generator=g, device=dev) / DIM ** 0.5
        nd = m // 100   # near-duplicates: a copy of another row of the block + 1 % noise
        src = torch.randint(0, m, (nd,), generator=g, device=dev)
        dst = torch.randint(0, m, (nd,), generator=g, device=dev)
        rows[dst] = rows[src] + 0.01 * torch.randn((nd, DIM), generator=g, device=dev) / DIM ** 0.5
        rows = torch.nn.functional.normalize(rows, dim=1)
        if first is None:
            first = rows[:4096].clone()
        index.add(rows)
    q = 2.0 * common + 0.5 * centres[torch.randint(0, 64, (nq,), generator=g, device=dev)] \
        + torch.randn((nq, DIM), generator=g, device=dev) / DIM ** 0.5
    planted = q[::5].shape[0]
    q[::5] = first[torch.randint(0, 4096, (planted,), generator=g, device=dev)] + 0.1 * torch.randn((planted, DIM), generator=g, device=dev) / DIM ** 0.5
    q = torch.nn.functional.normalize(q, dim=1)
    mean_cos = float((first[:1024] @ first[1024:2048].T).mean())
    ms, kernel_ms, status, s1, i1 = _time_search(index, q, K, steps, ev, lib, n, 0, True)
    ms_x, _, _, s2, i2 = _time_search(index, q, K, 2, ev, lib, n, 0, False)
    same = bool(torch.equal(i1, i2) and torch.equal(s1, s2))
    return {
        "value": round(nq / ms * 1e3, 1), "unit": "queries/s", "ms_per_step": round(ms, 4),
        "workload": f"{nq} queries x {n} rows: common component (mean pairwise cosine {mean_cos:.2f}), 64 clusters, 1 % near-duplicate "
                    f"rows, 20 % of the queries planted next to corpus rows; screened search (mean-centred bf16 copy)",
        "exact_fallback_queries": int(status[1]),
        "screen_kernel_ms": round(kernel_ms, 4),
        "equals_exact_scan_all_rows": same,
        "exact_scan_queries_per_s": round(nq / ms_x * 1e3, 1),
    }


def bench_search_cfg3(pkg, lib, dev, nq, ev, n_total: int = 8_841_823, steps: int = 3):
    """BASELINE cfg 3 sizes on ONE GPU: (a) the 1 105 228-row shard a rank of the 8-GPU job owns, screened (what the
    N = 8 step costs per rank before the all-gather); (b) the whole 8 841 823-row corpus, exact fp32 scan - the
    north star's "cosine top-k over 8.8M x 384-d" against the HBM roofline on SURVEY.md section 8(d)'s
    algorithmic bytes - and screened."""
    from semantic_search_kd_amd.dist import shard_bounds

    g = torch.Generator(device=dev).manual_seed(1234)
    index = pkg.FAISSIndexBuilder(embedding_dim=DIM, index_type="HNSW", metric="ip", device=str(dev))
    index.reserve(n_total)
    for lo in range(0, n_total, 1 << 20):
        rows = torch.randn((min(1 << 20, n_total - lo), DIM), generator=g, device=dev)
        index.add(torch.nn.functional.normalize(rows, dim=1))
    del rows
    q = torch.nn.functional.normalize(torch.randn((nq, DIM), generator=g, device=dev), dim=1)
    out = {}
    ms, kms, st, s1, i1 = _time_search(index, q, K, steps, ev, lib, n_total, 0, True)
    out["whole_screened"] = {"value": round(nq / ms * 1e3, 1), "unit": "queries/s", "ms_per_step": round(ms, 3), "rows": n_total,
                             "screen_kernel_ms": round(kms, 3), "exact_fallback_queries": int(st[1]),
                             "mfma_bf16_frac": round(2.0 * nq * n_total * DIM / (kms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)}
    ms, kms, _, s2, i2 = _time_search(index, q, K, 2, ev, lib, n_total, 0, False)
    qpb, passes, slices, waves, scans = (C.c_int() for _ in range(5))
    lib.sskd_index_search_plan(n_total, nq, K, qpb, passes, slices, waves, scans)
    alg = passes.value * n_total * DIM * 4 + nq * DIM * 4 + nq * K * 12
    out["whole_exact"] = {"value": round(nq / ms * 1e3, 1), "unit": "queries/s", "ms_per_step": round(ms, 3), "rows": n_total,
                          "scan_kernel_ms": round(kms, 3), "queries_per_block": qpb.value, "corpus_passes": passes.value,
                          "mfma_f32_frac": round(2.0 * nq * n_total * DIM / (kms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                          "hbm_algorithmic_bytes": alg,
                          "hbm_algorithmic_frac": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "equals_screened_all_rows": bool(torch.equal(i1, i2) and torch.equal(s1, s2))}
    del index, s1, i1, s2, i2
    torch.cuda.empty_cache()
    lo, hi = shard_bounds(n_total, 8, 0)
    g = torch.Generator(device=dev).manual_seed(1234)
    shard = pkg.FAISSIndexBuilder(embedding_dim=DIM, index_type="HNSW", metric="ip", device=str(dev), id_offset=lo)
    shard.add(torch.nn.functional.normalize(torch.randn((hi - lo, DIM), generator=g, device=dev), dim=1))
    ms, kms, st, _, _ = _time_search(shard, q, K, 5, ev, lib, hi - lo, lo, True)
    out["shard_of_8_screened"] = {"value": round(nq / ms * 1e3, 1), "unit": "queries/s", "ms_per_step": round(ms, 3), "rows": hi - lo,
                                  "screen_kernel_ms": round(kms, 3), "exact_fallback_queries": int(st[1])}
    return out


def bench_search_locality(pkg, lib, dev, n, nq, ev, steps: int = 5):
    """The search on a LOCALITY-ORDERED corpus (rows sorted by topic, a topic = documents of 8 adjacent chunks) with
    queries about documents - a tenth of them about the documents in the shard's first 2 048 rows, the rows the
    screening kernel's sample phase reads - at the bench batch size and at a batch large enough for ONE corpus slice
    (the geometry in which the whole sample lies inside the only slice).  Every output row is compared with the
    exact scan's (all queries, bit for bit): the shapes VERDICT r3 weak 1 named."""
    per_doc, topics = 8, 64
    g = torch.Generator(device=dev).manual_seed(2024)
    n_docs = n // per_doc
    centres = torch.nn.functional.normalize(torch.randn((topics, DIM), generator=g, device=dev), dim=1)
    index = pkg.FAISSIndexBuilder(embedding_dim=DIM, index_type="HNSW", metric="ip", device=str(dev))
    index.reserve(n_docs * per_doc)
    docs_per_topic = -(-n_docs // topics)
    doc_centres = []
    for lo in range(0, n_docs, 1 << 15):
        m = min(1 << 15, n_docs - lo)
        topic = (torch.arange(lo, lo + m, device=dev) // docs_per_topic)
        doc = torch.nn.functional.normalize(0.7 * centres[topic] + 0.7 * torch.nn.functional.normalize(
            torch.randn((m, DIM), generator=g, device=dev), dim=1), dim=1)
        doc_centres.append(doc)
        rows = 0.9 * doc.repeat_interleave(per_doc, dim=0) + 0.44 * torch.nn.functional.normalize(
            torch.randn((m * per_doc, DIM), generator=g, device=dev), dim=1)
        index.add(torch.nn.functional.normalize(rows, dim=1))
    doc_centres = torch.cat(doc_centres)
    rows_total = n_docs * per_doc
    out = {"workload": f"{rows_total} rows sorted by topic ({topics} topics, documents of {per_doc} adjacent chunks); every query is "
                       f"about one document, a tenth about the documents of the first 2 048 rows; screened search, all rows "
                       f"compared with the exact scan"}
    for name, batch in (("bench_batch", nq), ("large_batch", 40960)):
        which = torch.randint(0, n_docs, (batch,), generator=g, device=dev)
        which[::10] = torch.randint(0, 2048 // per_doc, (which[::10].shape[0],), generator=g, device=dev)
        q = torch.nn.functional.normalize(doc_centres[which] + 0.1 * torch.nn.functional.normalize(
            torch.randn((batch, DIM), generator=g, device=dev), dim=1), dim=1)
        qpb, passes, slices = C.c_int(), C.c_int(), C.c_int()
        lib.sskd_index_search_screened_plan(rows_total, batch, K, C.byref(qpb), C.byref(passes), C.byref(slices))
        ms, kernel_ms, status, s1, i1 = _time_search(index, q, K, steps, ev, lib, rows_total, 0, True)
        ms_x, _, _, s2, i2 = _time_search(index, q, K, 1, ev, lib, rows_total, 0, False)
        out[name] = {"value": round(batch / ms * 1e3, 1), "unit": "queries/s", "queries": batch, "ms_per_step": round(ms, 4),
                     "screen_kernel_ms": round(kernel_ms, 4), "queries_per_block": qpb.value, "slices": slices.value,
                     "exact_fallback_queries": int(status[1]),
                     "equals_exact_scan_all_rows": bool(torch.equal(i1, i2) and torch.equal(s1, s2)),
                     "mfma_bf16_frac": round(2.0 * batch * rows_total * DIM / (kernel_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)}
        del q, s1, i1, s2, i2
    return out


def self_launch(args) -> int:
    """``bench.py --gpus N`` from a bare shell: start the torch.distributed launcher as a child
    (never exec: this process must not have touched the GPU, and it has not) and return its code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--corpus", type=int, default=N_CORPUS)
    ap.add_argument("--queries", type=int, default=N_QUERIES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encode", action="store_true")
    ap.add_argument("--no-ragged", action="store_true", help="skip the ragged-length encode leg")
    ap.add_argument("--no-text", action="store_true", help="skip the text -> embedding leg")
    ap.add_argument("--exact-scan", action="store_true",
                    help="time the plain exact fp32 scan instead of the bf16-screened search (same results, bit for bit)")
    ap.add_argument("--no-train", action="store_true", help="skip the KD training-step leg (BASELINE cfg 4)")
    ap.add_argument("--no-teacher", action="store_true", help="skip the teacher cross-encoder leg (BASELINE cfg 5 model)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous + one all-gather over gloo on CPU, no GPU work: tests the N > 1 launch plumbing")
    ap.add_argument("--no-hostile", action="store_true", help="skip the anisotropic / near-duplicate search leg")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the BASELINE cfg-3 legs (8 841 823 rows: one shard + whole corpus)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        cpu_baselines_child()
        return

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))  # nothing above this line touches the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        import torch.distributed as dist

        assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            got = torch.empty(world, dtype=torch.int64)
            dist.all_gather_into_tensor(got, torch.tensor([rank], dtype=torch.int64))
            assert got.tolist() == list(range(world))
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "ranks": world}), flush=True)
        return
    # CPU baselines FIRST, in a child that never touches the GPU (rank 0 at N = 1 only); nothing above this line has
    # initialised the GPU in this process either
    cpu_lines = run_cpu_baselines_first(args) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        # RCCL builds its communicator lazily on the first collective: do that here, outside any
        # timed region (it is not part of a search step), whatever --warmup is
        _probe = torch.zeros(world * 4, device=dev)
        dist.all_gather_into_tensor(_probe, torch.ones(4, device=dev))
        torch.cuda.synchronize()
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import semantic_search_kd_amd as pkg
    from semantic_search_kd_amd import _native
    from semantic_search_kd_amd.dist import ShardedSearcher, shard_bounds

    lib = _native.load()
    n, nq = args.corpus, args.queries

    # ---- synthetic inputs, generated on device ---------------------------
    lo, hi = shard_bounds(n, world, rank)
    gen = torch.Generator(device=dev).manual_seed(1234)
    shard_chunks = []
    # every rank draws the same stream and keeps its rows, so the global corpus is independent of N
    for c_lo in range(0, n, 1 << 18):
        c_hi = min(c_lo + (1 << 18), n)
        block = torch.randn((c_hi - c_lo, DIM), generator=gen, device=dev, dtype=torch.float32)
        s_lo, s_hi = max(lo, c_lo), min(hi, c_hi)
        if s_lo < s_hi:
            shard_chunks.append(block[s_lo - c_lo : s_hi - c_lo].clone())
        del block
    shard = torch.cat(shard_chunks) if shard_chunks else torch.empty((0, DIM), device=dev)
    del shard_chunks
    shard /= shard.norm(dim=1, keepdim=True)
    qgen = torch.Generator(device=dev).manual_seed(4321)
    queries = torch.randn((nq, DIM), generator=qgen, device=dev, dtype=torch.float32)
    queries /= queries.norm(dim=1, keepdim=True)

    index = pkg.FAISSIndexBuilder(embedding_dim=DIM, index_type="HNSW", metric="ip", device=str(dev), id_offset=lo)
    index.add(shard)
    screened_bytes = int(lib.sskd_index_search_screened_workspace_bytes(index.ntotal, nq, K))
    use_screen = not args.exact_scan and screened_bytes > 0
    index.screening = use_screen
    n_local = index.ntotal
    def local_search(q, k, out_scores=None, out_ids=None):
        return index.search_device(q, k, normalize_queries=False, out_scores=out_scores, out_ids=out_ids)

    searcher = ShardedSearcher(local_search)

    # profiled variant of the local scan (events around the scan kernel, same stream)
    ev = HipEvents()
    ws_bytes = screened_bytes if use_screen else int(lib.sskd_index_search_workspace_bytes(n_local, nq, K))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    if use_screen:
        index.search_device(queries[:64], K, normalize_queries=False)  # builds the bf16 screening copy
    out_s = torch.empty((nq, K), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, K), dtype=torch.int64, device=dev)
    ev_pairs = [(ev.create(), ev.create()) for _ in range(args.steps)]

    def local_search_profiled(step, out_scores=None, out_ids=None):
        a, b = ev_pairs[step]
        o_s = out_s if out_scores is None else out_scores
        o_i = out_i if out_ids is None else out_ids
        st_ptr = int(torch.cuda.current_stream(dev).cuda_stream)
        if use_screen:
            _native.check(
                lib.sskd_index_search_screened(
                    index._tiled.data_ptr(), index._bf16.data_ptr(), n_local, queries.data_ptr(), nq, K, lo,
                    o_s.data_ptr(), o_i.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), st_ptr, a, b,
                )
            )
        else:
            _native.check(
                lib.sskd_index_search_profiled(
                    index._tiled.data_ptr(), n_local, queries.data_ptr(), nq, K, lo,
                    o_s.data_ptr(), o_i.data_ptr(), ws.data_ptr(), ws.numel(), st_ptr, a, b,
                )
            )
        return o_s, o_i

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = searcher.search(queries, K)
    barrier()
    t0 = time.perf_counter()
    for step in range(args.steps):
        searcher.local_search = (lambda q, k, out_scores=None, out_ids=None, _s=step:
                                 local_search_profiled(_s, out_scores, out_ids))
        res = searcher.search(queries, K)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    qps = nq * args.steps / dt

    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # parity spot check of what was just measured: first 100 queries against the reference idiom
        # (numpy sgemm + argsort) on the host, same vectors
        from oracle import search as oracle

        corpus_host, queries_host = shard.cpu().numpy(), queries[:100].cpu().numpy()
        ref_s, ref_i = oracle.topk_blas(queries_host, corpus_host, K)
        del corpus_host
        got_s, got_i = res[0][:100].cpu().numpy(), res[1][:100].cpu().numpy()
        ties = set(oracle.near_tie_queries(ref_s).tolist())
        ok = all(np.array_equal(got_i[j], ref_i[j]) for j in range(100) if j not in ties)
        parity = {"checked_queries": 100, "ids_identical": bool(ok),
                  "max_abs_score_diff": float(np.abs(got_s - ref_s).max()), "near_tie_queries": len(ties)}

    scan_ms = float(np.mean([ev.elapsed_ms(a, b) for a, b in ev_pairs]))
    flops = 2.0 * nq * n_local * DIM
    traffic, traffic_from = None, None
    if use_screen:
        # screening kernel: bf16 MFMA (32x32x16) over a bf16 copy of the rows; B_q = 128 queries per workgroup
        assert int(status[0].item()) == 0, "screened search overflowed its exact fallback"
        qpb, passes, slices = (C.c_int() for _ in range(3))
        _native.check(lib.sskd_index_search_screened_plan(n_local, nq, K, qpb, passes, slices))
        qpb_v, passes_v, slices_v = qpb.value, passes.value, slices.value
        alg_bytes = passes_v * n_local * DIM * 2 + nq * DIM * 4 + nq * K * 12
        kernel_name, peak_tf = "screen_append_kernel", MFMA_BF16_PEAK_TF
        tpath = REPO / "profiles" / "screen_traffic.json"
    else:
        qpb, passes, slices, waves, scans = (C.c_int() for _ in range(5))
        lib.sskd_index_search_plan(n_local, nq, K, qpb, passes, slices, waves, scans)
        qpb_v, passes_v, slices_v = qpb.value, passes.value, slices.value
        # algorithmic bytes of one scan launch: P passes x rows x 1536 B + queries + partial lists
        alg_bytes = passes_v * n_local * DIM * 4 + nq * DIM * 4 + nq * K * 12
        kernel_name, peak_tf = "scan_topk_kernel", MFMA_F32_PEAK_TF
        tpath = REPO / "profiles" / "scan_traffic.json"
    achieved_gbs = alg_bytes / (scan_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters: they cannot be read from inside this process, so the number comes
    # from the committed rocprofv3 --pmc passes of this same command - and only while search.hip is byte-identical to
    # the file those passes ran (sha recorded beside the number); otherwise traffic is null and says why
    traffic_note = None
    if tpath.exists() and world == 1 and n == N_CORPUS and nq == N_QUERIES:
        try:
            tj = json.loads(tpath.read_text())
            if tj.get("search_hip_sha") == search_hip_sha():
                traffic, traffic_from = tj.get("hbm_bytes_per_launch"), tj.get("from")
            else:
                traffic_note = f"stale: counters were collected at search.hip {tj.get('search_hip_sha')}, this is {search_hip_sha()}"
        except Exception:
            traffic = None
    line = {
        "metric": METRIC,
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "rccl_ranks": dist.get_world_size() if world > 1 else 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if not use_screen else "bf16 screen + f32 exact",
        "data": "synthetic",
        "config": {
            "workload": f"cosine top-{K}: {nq} queries x {n} x {DIM}-d fp32 corpus in HBM (configs[1] search), "
            f"row-sharded over {world} GPU(s), all-gather + merge",
            "corpus_rows": n,
            "queries": nq,
            "k": K,
            "queries_per_block": qpb_v,
            "corpus_passes": passes_v,
            "slices": slices_v,
            "search_path": "bf16-screened + exact fp32 re-scoring (bit-identical to the exact scan)" if use_screen
            else "exact fp32 scan",
            "exact_fallback_queries": int(status[1].item()) if use_screen else 0,
            # HBM held by this rank's shard: the fp32 index (one row-major copy: scan, re-scoring, save), the screening
            # sidecar (bf16 tiles + norms) and the per-call workspace of the timed path
            "index_bytes": int(lib.sskd_index_tiled_bytes(n_local)),
            "sidecar_bytes": int(lib.sskd_index_bf16_bytes(n_local)) if use_screen else 0,
            "workspace_bytes": int(ws_bytes),
        },
        # roofline of the dominant kernel.  Screened: bf16 MFMA peak (the B_q-dependent "algorithmic HBM bytes" of
        # SURVEY.md section 8(d) are NOT reported for it - with 128 queries per workgroup they exceed what HBM can move,
        # the slices of an XCD share tiles through L2; the counter traffic is the honest HBM figure).  Exact scan:
        # fp32 MFMA peak, plus the section 8(d) algorithmic bytes against the 8 TB/s HBM peak.
        "roofline": {
            "bound": "mfma",
            "kernel": kernel_name,
            "achieved": round(flops / (scan_ms * 1e-3) / 1e12, 2),
            "peak": peak_tf,
            "unit": "TFLOP/s",
            "frac": round(flops / (scan_ms * 1e-3) / 1e12 / peak_tf, 4),
            "traffic": traffic,
            "traffic_from": traffic_from,
            "traffic_note": traffic_note,
            "hbm_counter_gbs": round(traffic / (scan_ms * 1e-3) / 1e9, 1) if traffic else None,
            "kernel_ms": round(scan_ms, 4),
            "algorithmic_flops": flops,
        },
    }
    if not use_screen:
        line["roofline"].update({
            "hbm_algorithmic_bytes": alg_bytes,
            "hbm_algorithmic_gbs": round(achieved_gbs, 1),
            "hbm_algorithmic_frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
            "hbm_peak_gbs": HBM_PEAK_GBS,
        })

    # ---- hostile-data leg and the cfg-3 sizes (rank 0, N = 1) -----------------------------------------------------
    if rank == 0 and world == 1 and not args.no_hostile and n >= 100_000:
        line["search_anisotropic"] = bench_search_anisotropic(pkg, lib, dev, n, nq, ev)
        line["search_locality"] = bench_search_locality(pkg, lib, dev, n, nq, ev)
    if rank == 0 and world == 1 and not args.no_cfg3:
        del index, shard
        torch.cuda.empty_cache()
        line["search_cfg3"] = bench_search_cfg3(pkg, lib, dev, nq, ev)

    # ---- encoder leg (docs embedded / s), once the kernels exist ---------------------------
    if not args.no_encode:
        try:
            from semantic_search_kd_amd import bench_encode  # noqa: F401
        except ImportError:
            bench_encode = None
        if bench_encode is not None:
            line["encode"] = bench_encode(dev, world, args.steps, args.warmup, barrier,
                                          ragged=not args.no_ragged and rank == 0,
                                          text=not args.no_text and rank == 0)
    # ---- teacher cross-encoder (cfg 5 model, data parallel) and KD training step (cfg 4) -------
    if not args.no_teacher:
        from semantic_search_kd_amd.bench_support import bench_teacher

        line["teacher"] = bench_teacher(dev, world, max(2, args.steps // 2), 1, barrier)
    if rank == 0 and world == 1 and not args.no_train:
        from semantic_search_kd_amd.bench_support import bench_kd_step

        line["kd_step"] = bench_kd_step(dev)

    # ---- CPU baselines (measured first, in the child: see run_cpu_baselines_first) + parity spot check ----------
    if cpu_lines is not None:
        if "search" in cpu_lines:
            line["cpu_baseline"] = cpu_lines["search"]
        if "encode" in cpu_lines and "encode" in line:
            line["encode"]["cpu_baseline"] = cpu_lines["encode"]
        if "error" in cpu_lines:
            line["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": physical_cores(), "kind": "port",
                                    "sample": "child process failed: " + cpu_lines["error"]}
    if parity is not None:
        line["parity"] = parity

    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
