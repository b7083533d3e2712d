#!/usr/bin/env python3
"""Build a search index from a parquet corpus with the MI355X encoder.

Same flags as the reference's ``scripts/build_faiss_index.py:15-24`` (``--hnsw-*`` are accepted and
ignored: the index is an exact scan, there is no graph to build).
Run as ``python -m semantic_search_kd_amd.build_index_cli ...``.

Row-sharded build (BASELINE cfg 3): launch the same command under
``python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 -m semantic_search_kd_amd.build_index_cli ...``
and rank r encodes rows ``shard_bounds(N, G, r)`` straight into its own HBM shard and writes
``<output-dir>/shard_<r>/`` (+ ``shards.json`` from rank 0): see ``sharded_index.py``.  ``--shards 1`` forces the
manifest layout from a single process.
"""
from __future__ import annotations

import argparse
import os
import re
import sys
from pathlib import Path

from .index import FAISSIndexBuilder
from .student import StudentModel


def _positive(value: str) -> int:
    v = int(value)
    if v <= 0:
        raise argparse.ArgumentTypeError(f"must be a positive integer, got {value}")
    return v


def _device(value: str) -> str:
    # the reference accepts cpu | cuda | cuda:N (scripts/_validate_args.py:35-39); this backend is GPU-only
    if not re.fullmatch(r"cuda(:\d+)?", value):
        raise argparse.ArgumentTypeError(f"--device must be cuda or cuda:N (MI355X backend, no CPU path), got {value!r}")
    return value


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--model-path", type=str, required=True, help="local model directory")
    ap.add_argument("--data-path", type=str, required=True, help="parquet corpus (columns: text, chunk_id)")
    ap.add_argument("--output-dir", type=str, required=True)
    ap.add_argument("--max-docs", type=_positive, default=None)
    ap.add_argument("--batch-size", type=_positive, default=32)
    ap.add_argument("--device", type=_device, default="cuda")
    ap.add_argument("--hnsw-m", type=_positive, default=32, help="accepted for compatibility; unused")
    ap.add_argument("--hnsw-ef-construction", type=_positive, default=200, help="accepted for compatibility; unused")
    ap.add_argument("--shards", type=int, default=0,
                    help="0 = one shard per rank when launched under torch.distributed.run, plain single index otherwise; "
                         "1 = write the sharded layout (shards.json + shard_0/) from this single process")
    args = ap.parse_args(argv)
    for flag, p in (("--model-path", args.model_path), ("--data-path", args.data_path)):
        if not Path(p).exists():
            ap.error(f"{flag}: {p} does not exist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.shards == 1:
        return _build_sharded(args, world)

    model = StudentModel(args.model_path, device=args.device)
    builder = FAISSIndexBuilder(embedding_dim=384, index_type="HNSW", metric="cosine", device=args.device)
    index = builder.build_from_parquet(
        model=model,
        parquet_path=Path(args.data_path),
        batch_size=args.batch_size,
        max_docs=args.max_docs,
        hnsw_m=args.hnsw_m,
        hnsw_ef_construction=args.hnsw_ef_construction,
    )
    builder.save(Path(args.output_dir))
    print(f"Index saved to: {args.output_dir}")
    print(f"Total vectors: {index.ntotal}")
    return 0


def _build_sharded(args, world: int) -> int:
    """One process per GPU (``torch.distributed.run`` sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    import torch
    import torch.distributed as dist

    from .sharded_index import build_sharded

    device = args.device
    started = False
    if world > 1:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        one_gpu_each = torch.cuda.device_count() >= world
        device = f"cuda:{local_rank}" if one_gpu_each else args.device   # rehearsal: ranks share a device over gloo
        torch.cuda.set_device(torch.device(device if ":" in device else "cuda:0"))
        if not dist.is_initialized():
            if one_gpu_each:
                dist.init_process_group("nccl", device_id=torch.device(device))   # RCCL over xGMI
            else:
                dist.init_process_group("gloo")
            started = True
    try:
        model = StudentModel(args.model_path, device=device)
        manifest = build_sharded(model, Path(args.data_path), Path(args.output_dir), batch_size=args.batch_size,
                                 max_docs=args.max_docs, device=device)
        if not dist.is_initialized() or dist.get_rank() == 0:
            print(f"Index saved to: {args.output_dir}")
            print(f"Total vectors: {manifest['n_total']} in {len(manifest['shards'])} shard(s)")
    finally:
        if started:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
