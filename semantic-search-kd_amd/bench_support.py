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


def bench_encode(device, world: int, steps: int, warmup: int, barrier, batch: int = 512, seq_len: int = 256,
                 ragged: bool = True, text: bool = False):
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
        # rank 0's own rate on MS MARCO-shaped ragged lengths (per GPU, not aggregated)
        "ragged": bench_encode_ragged(enc, device) if ragged else None,
    }


def marco_like_lengths(n: int, seed: int = 7, max_len: int = 256) -> np.ndarray:
    """Token lengths of an MS MARCO-shaped passage set (clipped log-normal, mean ~75,
    median ~68, max 256 - an assumption of the survey, not pinned by the reference)."""
    rng = np.random.default_rng(seed)
    lens = rng.lognormal(mean=np.log(68.0), sigma=0.45, size=n)
    return np.clip(np.rint(lens), 8, max_len).astype(np.int64)


def bench_encode_ragged(enc: Mi355xSentenceEncoder, device, passes: int = 2, n_docs: int = 32768, batch: int = 512):
    """docs/s on ragged lengths the way ``encode()`` runs them: sort by length (longest first),
    batches of ``batch``, each padded to its longest member (rounded up to the 32-token tile).
    FLOPs are counted on real tokens only.  (The sample must be large against the batch size: with
    only a few batches each one spans a wide range of lengths and the padding share is inflated.
    Cutting batches at tile-count boundaries instead was measured: no gain at this size.)"""
    cfg = enc.config
    lens = np.sort(marco_like_lengths(n_docs))[::-1]
    g = torch.Generator(device=device).manual_seed(11)
    batches = []
    flops = 0.0
    for b0 in range(0, n_docs, batch):
        bl = lens[b0:b0 + batch]
        S = int(-(-int(bl[0]) // 32) * 32)
        ids = torch.randint(999, cfg.vocab_size, (len(bl), S), generator=g, device=device, dtype=torch.int32)
        ids[:, 0] = 101
        lt = torch.from_numpy(bl.copy()).to(device)
        mask = (torch.arange(S, device=device)[None, :] < lt[:, None]).to(torch.int32)
        ids = ids * mask
        batches.append((ids, mask, torch.empty((len(bl), cfg.hidden_size), dtype=torch.float32, device=device)))
        flops += float(sum(encoder_flops(int(l), int(l), cfg) for l in bl))
    def one_pass():
        for ids, mask, out in batches:
            enc.encode_token_ids(ids, mask, normalize=True, out=out)
    one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        one_pass()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / passes
    padded = sum(int(i.numel()) for i, _, _ in batches)
    return {
        "value": round(n_docs / dt, 1),
        "unit": "docs/s",
        "workload": f"{n_docs} passages, lengths clipped log-normal (mean {lens.mean():.0f}, median "
                    f"{int(np.median(lens))}, max {int(lens.max())} tokens; SURVEY.md §8d assumption), "
                    f"length-sorted batches of {batch} padded to the longest member",
        "real_tokens_per_s": round(float(lens.sum()) / dt, 1),
        "padding_overhead": round(padded / float(lens.sum()) - 1.0, 4),
        "mfma_frac_real_tokens": round(flops / dt / 1e12 / MFMA_BF16_PEAK_TF, 4),
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
