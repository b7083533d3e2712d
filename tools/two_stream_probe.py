"""Same-process A/B: 512 x 256 encode as one launch chain vs two half-batches on two HIP streams
(`Mi355xSentenceEncoder.split_streams`), and the ragged packed path with / without alternating streams."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd.bench_support import marco_like_lengths, synthetic_ids  # noqa: E402
from semantic_search_kd_amd.encoder import Mi355xSentenceEncoder  # noqa: E402
from semantic_search_kd_amd.weights import BertConfig  # noqa: E402

dev = torch.device("cuda:0")
cfg = BertConfig()
enc = Mi355xSentenceEncoder.from_synthetic(cfg, device=str(dev))
ids, mask = synthetic_ids(512, 256, cfg.vocab_size, dev)
out = torch.empty((512, 384), device=dev)
lens = marco_like_lengths(32768).astype(np.int32)
flat = np.random.default_rng(1).integers(999, cfg.vocab_size, size=int(lens.sum())).astype(np.int32)
out_r = torch.empty((32768, 384), device=dev)


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rnd in range(3):
    for split in (False, True):
        enc.split_streams = split
        a = timed(lambda: enc.encode_token_ids(ids, mask, out=out), 10)
        b = timed(lambda: enc.encode_ragged(flat, lens, out=out_r), 2)
        print(f"round {rnd} split_streams={split}: 512x256 {a:.3f} ms; ragged 32768 docs {b:.1f} ms ({32768 / b * 1e3:.0f} docs/s)", flush=True)
