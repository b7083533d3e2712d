"""Serve a row-sharded index from the GPUs of one node.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m semantic_search_kd_amd.serve.sharded --model-path M --index-dir D [--host H --port P]

Rank 0 runs the FastAPI application of ``serve/app.py`` (the reference's surface: src/serve/app.py:221-457) with a
``ShardedIndex`` as its index object; ranks 1 .. G-1 hold their shards in HBM and answer rank 0's searches
(``ShardedIndex.serve_forever``: broadcast of the query block, local scan, ONE all-gather, merge).  The student
model lives on rank 0 only - a query is encoded once.  Without ``torch.distributed.run`` this is a one-process server
of the same directory (all shards on one GPU).
"""
from __future__ import annotations

import argparse
import datetime
import os
import sys

import torch


def init_group_from_env() -> tuple:
    """``(world, rank, device)``; starts the process group when launched under torch.distributed.run."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 1, 0, "cuda"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu_each = torch.cuda.device_count() >= world
    device = f"cuda:{local_rank}" if one_gpu_each else "cuda:0"
    torch.cuda.set_device(torch.device(device))
    if not dist.is_initialized():
        # The DATA group's timeout bounds one search (query broadcast + all-gather, entered only after every rank
        # has reported a successful local scan).  Idle ranks do not wait in it: they park in a host broadcast on
        # ShardedIndex's gloo control group, which no watchdog aborts (sharded_index.py, "Failure path").
        timeout = datetime.timedelta(seconds=float(os.environ.get("SEMANTIC_KD_SHARD_OP_TIMEOUT_S", "120")))
        if one_gpu_each:
            dist.init_process_group("nccl", device_id=torch.device(device), timeout=timeout)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo", timeout=timeout)                  # rehearsal: ranks share a GPU
    return world, dist.get_rank(), device


def main(argv=None) -> int:
    import torch.distributed as dist

    from ..sharded_index import ShardedIndex

    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--index-dir", required=True)
    ap.add_argument("--teacher-path", default=None)
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    args = ap.parse_args(argv)
    world, rank, device = init_group_from_env()
    index = ShardedIndex(device=device, op_timeout_s=float(os.environ.get("SEMANTIC_KD_SHARD_OP_TIMEOUT_S", "120")))
    try:
        index.load_all_ranks(args.index_dir)   # ShardFailure on EVERY rank when any rank cannot load its shards
    except Exception as exc:  # noqa: BLE001
        print(f"[rank {rank}] cannot serve {args.index_dir}: {exc}", file=sys.stderr, flush=True)
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()
        return 1
    try:
        if rank != 0:
            index.serve_forever()
            return 0
        import uvicorn

        from .app import app_state, create_app

        app = create_app(student_model_path=args.model_path, teacher_model_path=args.teacher_path, device=device)
        app_state.index_builder, app_state.doc_ids, app_state.doc_texts = index, index.doc_ids, index.doc_texts
        try:
            uvicorn.run(app, host=args.host, port=args.port)
        finally:
            index.close()
        return 0
    finally:
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
