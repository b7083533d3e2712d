"""Where the text -> embedding path spends its time: ``python tools/text_path_probe.py``.
Sweeps the tokenise-chunk size of ``Mi355xSentenceEncoder.encode`` and times the stages alone."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from semantic_search_kd_amd import encoder as enc_mod  # noqa: E402
from semantic_search_kd_amd.bench_support import synthetic_passages, synthetic_vocab  # noqa: E402
from semantic_search_kd_amd.encoder import Mi355xSentenceEncoder, build_wordpiece_tokenizer  # noqa: E402
from semantic_search_kd_amd.student import StudentModel  # noqa: E402
from semantic_search_kd_amd.weights import BertConfig  # noqa: E402

dev = torch.device("cuda:0")
cfg = BertConfig()
enc = Mi355xSentenceEncoder.from_synthetic(cfg, device="cuda:0")
vocab = synthetic_vocab(cfg.vocab_size)
enc.tokenizer = build_wordpiece_tokenizer(vocab)
student = StudentModel.from_encoder(enc, "probe")
N = 32768
docs = synthetic_passages(vocab, N)
student.encode_documents(docs[:4096], batch_size=32)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


pre = ["passage: " + d for d in docs]
t_tok = timed(lambda: enc._tokenize_flat(pre))
flat, lengths = enc._tokenize_flat(pre)
out = torch.empty((N, cfg.hidden_size), dtype=torch.float32, device=dev)
t_rag = timed(lambda: enc.encode_ragged(flat, lengths, normalize=True, out=out))
t_d2h = timed(lambda: out.cpu().numpy())
print(f"tokenise alone {t_tok * 1e3:.1f} ms ({N / t_tok:.0f}/s); encode_ragged alone {t_rag * 1e3:.1f} ms ({N / t_rag:.0f}/s); "
      f"D2H {t_d2h * 1e3:.1f} ms", flush=True)
for chunk in (2048, 4096, 8192, 16384, 32768):
    enc_mod.TOKENIZE_CHUNK = chunk
    t = timed(lambda: student.encode_documents(docs, batch_size=32))
    print(f"chunk {chunk:6d}: {t * 1e3:7.1f} ms  {N / t:9.0f} docs/s", flush=True)
