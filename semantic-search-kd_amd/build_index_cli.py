#!/usr/bin/env python3
"""Build a search index from a parquet corpus with the MI355X encoder.

Same flags as the reference's ``scripts/build_faiss_index.py:15-24`` (``--hnsw-*`` are accepted and
ignored: the index is an exact scan, there is no graph to build).
Run as ``python -m semantic_search_kd_amd.build_index_cli ...``.
"""
from __future__ import annotations

import argparse
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
    args = ap.parse_args(argv)
    for flag, p in (("--model-path", args.model_path), ("--data-path", args.data_path)):
        if not Path(p).exists():
            ap.error(f"{flag}: {p} does not exist")

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


if __name__ == "__main__":
    sys.exit(main())
