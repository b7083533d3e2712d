"""Encoder legs of ``bench.py`` and ``__graft_entry__.smoke()`` (product code only: no oracle here)."""
from __future__ import annotations

import time

import numpy as np
import torch

from .encoder import Mi355xSentenceEncoder
from .weights import BertConfig

MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16


def encoder_flops(tokens: int, seq_len: int, cfg: BertConfig) -> float:
    """Algorithmic FLOPs of the forward pass over real tokens:
    per token 12 * (2 * (4 H^2 + 2 H F) + 4 S H)."""
    h, f = cfg.hidden_size, cfg.intermediate_size
    return tokens * cfg.num_hidden_layers * (2.0 * (4 * h * h + 2 * h * f) + 4.0 * seq_len * h)


def synthetic_ids(batch: int, seq_len: int, vocab: int, device, seed: int = 0):
    """ids uniform in [999, vocab), [CLS] first, [SEP] last, full mask."""
    g = torch.Generator(device=device).manual_seed(seed)
    ids = torch.randint(999, vocab, (batch, seq_len), generator=g, device=device, dtype=torch.int32)
    ids[:, 0] = 101
    ids[:, -1] = 102
    return ids, torch.ones_like(ids)


def bench_encode(device, world: int, steps: int, warmup: int, barrier, batch: int = 512, seq_len: int = 256):
    """docs embedded / s: every rank encodes its own ``batch x seq_len`` synthetic batches
    (pure data parallel, no communication); whole-job rate = world * batch * steps / max-rank time."""
    import torch.distributed as dist

    cfg = BertConfig()
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device=str(device))
    ids, mask = synthetic_ids(batch, seq_len, cfg.vocab_size, device, seed=int(device.index or 0))
    out = torch.empty((batch, cfg.hidden_size), dtype=torch.float32, device=device)
    for _ in range(max(warmup, 1)):
        enc.encode_token_ids(ids, mask, normalize=True, out=out)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        enc.encode_token_ids(ids, mask, normalize=True, out=out)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    docs_per_s = world * batch * steps / dt
    flops = encoder_flops(batch * seq_len, seq_len, cfg)
    tf_per_gpu = flops * steps / dt / 1e12
    norms = out.norm(dim=1)
    return {
        "value": round(docs_per_s, 1),
        "unit": "docs/s",
        "ms_per_step": round(dt / steps * 1e3, 4),
        "dtype": "bf16",
        "config": {"workload": f"e5-small-v2-shaped encoder, batch {batch} x seq_len {seq_len} per GPU, "
                               "synthetic ids, random-init weights", "layers": cfg.num_hidden_layers},
        "roofline": {
            "bound": "mfma",
            "achieved": round(tf_per_gpu, 1),
            "peak": MFMA_BF16_PEAK_TF,
            "unit": "TFLOP/s",
            "frac": round(tf_per_gpu / MFMA_BF16_PEAK_TF, 4),
            "algorithmic_flops_per_step": flops,
        },
        "unit_norm_ok": bool(torch.allclose(norms, torch.ones_like(norms), atol=1e-3)),
    }


def encoder_smoke_embeddings(device: str = "cuda:0"):
    """Tiny forward of a 2-layer synthetic encoder on the golden l2 input; returns (embeddings, ids, mask)
    for the caller (``__graft_entry__.smoke``) to compare against the committed golden vectors."""
    from pathlib import Path

    golden = np.load(Path(__file__).resolve().parent.parent / "tests" / "golden" / "bert_l2.npz")
    cfg = BertConfig(num_hidden_layers=int(golden["layers"]))
    enc = Mi355xSentenceEncoder.from_synthetic(cfg, device=device)
    emb = enc.encode_token_ids(golden["input_ids"], golden["attention_mask"], normalize=True)
    torch.cuda.synchronize()
    return emb.cpu().numpy(), golden["embeddings"]
