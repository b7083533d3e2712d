"""Diagnostic: latency of encoding small batches of short sequences (the online /search shape)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd.bench_support import synthetic_ids  # noqa: E402
from semantic_search_kd_amd.encoder import Mi355xSentenceEncoder  # noqa: E402
from semantic_search_kd_amd.weights import BertConfig  # noqa: E402

dev = torch.device("cuda:0")
cfg = BertConfig()
enc = Mi355xSentenceEncoder.from_synthetic(cfg, device=str(dev))
for B, S in [(1, 32), (1, 64), (8, 32), (32, 32), (32, 128), (128, 64)]:
    ids, mask = synthetic_ids(B, S, cfg.vocab_size, dev)
    out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device=dev)
    for _ in range(5):
        enc.encode_token_ids(ids, mask, normalize=True, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        enc.encode_token_ids(ids, mask, normalize=True, out=out)
    torch.cuda.synchronize()
    print(f"B={B} S={S}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call", flush=True)
