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


def cpu_search_baseline(corpus_host: np.ndarray, queries_host: np.ndarray, k: int, budget_s: float = 20.0):
    """The reference's CPU exact-search idiom on the box's host cores (kind "port"):
    ``np.matmul(q, corpus.T)`` + top-k by argsort (scripts/simple_eval.py:25,35), chunked over
    the corpus so the score matrix stays small.  Bounded sample: as many 100-query batches as
    fit ~budget_s."""
    from oracle import search as oracle  # checker / baseline leg only

    done, t0 = 0, time.perf_counter()
    batch = 100
    while done < queries_host.shape[0]:
        q = queries_host[done : done + batch]
        best_s = np.full((q.shape[0], k), -np.inf, np.float32)
        best_i = np.full((q.shape[0], k), -1, np.int64)
        for lo in range(0, corpus_host.shape[0], 131072):
            s = oracle.scores_blas(q, corpus_host[lo : lo + 131072])
            part = np.argpartition(-s, k - 1, axis=1)[:, :k]
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
    done, t0 = 0, time.perf_counter()
    while done < max_docs:
        e = one_batch()
        done += batch
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    assert np.isfinite(e).all()
    return {
        "value": round(done / dt, 2),
        "unit": "docs/s",
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
    args = ap.parse_args()

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
        kernel_name, peak_tf = "screen_topk_kernel", MFMA_BF16_PEAK_TF
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
    # HBM bytes per launch from the PMC counters: they cannot be read from inside this process, so the
    # number comes from the committed rocprofv3 --pmc passes of this same command and says so
    if tpath.exists() and world == 1 and n == N_CORPUS and nq == N_QUERIES:
        try:
            tj = json.loads(tpath.read_text())
            traffic, traffic_from = tj.get("hbm_bytes_per_launch"), tj.get("from")
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
        },
        # The scan is bound by the fp32 matrix pipe, not by HBM: with B_q = 64 queries per block its
        # MFMAs saturate at 4.9 TB/s of *algorithmic* corpus traffic, and the XCD-affine slice mapping
        # serves most re-reads from L2 (see "traffic").  Both views are reported; "hbm_*" uses the
        # B_q-dependent algorithmic bytes of SURVEY.md §8(d) against the 8 TB/s HBM peak.
        "roofline": {
            "bound": "mfma",
            "kernel": kernel_name,
            "achieved": round(flops / (scan_ms * 1e-3) / 1e12, 2),
            "peak": peak_tf,
            "unit": "TFLOP/s",
            "frac": round(flops / (scan_ms * 1e-3) / 1e12 / peak_tf, 4),
            "traffic": traffic,
            "traffic_from": traffic_from,
            "kernel_ms": round(scan_ms, 4),
            "algorithmic_flops": flops,
            "hbm_algorithmic_bytes": alg_bytes,
            "hbm_algorithmic_gbs": round(achieved_gbs, 1),
            "hbm_algorithmic_frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
            "hbm_peak_gbs": HBM_PEAK_GBS,
        },
    }

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

    # ---- CPU baselines: rank 0, N = 1 only, AFTER every GPU leg (their BLAS / OpenMP worker threads keep
    # spinning for a while and slow the launch thread of whatever GPU leg follows: the KD step read 41 ms
    # instead of 35 ms when the encoder's CPU baseline ran before it) ---------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and "encode" in line:
        line["encode"]["cpu_baseline"] = cpu_encode_baseline()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        corpus_host = shard.cpu().numpy()
        queries_host = queries.cpu().numpy()
        cpu_qps, cpu_nq, cpu_dt = cpu_search_baseline(corpus_host, queries_host, K)
        threads = torch.get_num_threads()
        try:
            from threadpoolctl import threadpool_info

            blas = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
            threads = max(blas) if blas else threads
        except Exception:
            pass
        line["cpu_baseline"] = {
            "value": round(cpu_qps, 2),
            "unit": "queries/s",
            "cores": threads,
            "kind": "port",
            "sample": f"{cpu_nq} of the {nq} queries against the full {n}-row corpus "
            f"(numpy sgemm + argpartition, the reference's exact-search idiom), {cpu_dt:.1f} s",
            "host_cpus": os.cpu_count(),
            "faiss_hnsw": faiss_probe(),
        }
        # parity spot check of what was just measured (first 100 queries, vs the same idiom)
        from oracle import search as oracle

        ref_s, ref_i = oracle.topk_blas(queries_host[:100], corpus_host, K)
        got_s, got_i = res[0][:100].cpu().numpy(), res[1][:100].cpu().numpy()
        ties = set(oracle.near_tie_queries(ref_s).tolist())
        ok = all(np.array_equal(got_i[j], ref_i[j]) for j in range(100) if j not in ties)
        line["parity"] = {
            "checked_queries": 100,
            "ids_identical": bool(ok),
            "max_abs_score_diff": float(np.abs(got_s - ref_s).max()),
            "near_tie_queries": len(ties),
        }

    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
