"""Teacher cross-encoder oracle (CPU, fp32) - TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``transformers.XLMRobertaForSequenceClassification`` with one label - the model behind the
reference's ``TeacherModel`` / ``CrossEncoder.predict`` (src/mining/miners.py:135-137,
src/serve/app.py:325-326; bge-reranker-large, docs/adr-002):

    x      = post-LN BERT encoder of oracle/encoder.py with RoBERTa position ids (padding_idx + 1 + t)
    logit  = out_proj( tanh( dense( x[:, 0] ) ) )

Pinned: tests/golden/make_golden.py asserts it equal to the transformers class built from an in-memory
config on synthetic weights and commits the logits (tests/golden/xlmr_small.npz).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import encoder as enc


def logits(sd: Dict[str, np.ndarray], input_ids: np.ndarray, attention_mask: np.ndarray, num_layers: int,
           num_heads: int, eps: float = 1e-5, pad_token_id: int = 1) -> np.ndarray:
    t = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in sd.items()}
    h = enc.bert_hidden_states_torch(t, input_ids, attention_mask, num_layers, num_heads, eps,
                                     pos_offset=pad_token_id + 1)[-1]
    x = torch.tanh(h[:, 0] @ t["classifier.dense.weight"].T + t["classifier.dense.bias"])
    return (x @ t["classifier.out_proj.weight"].T + t["classifier.out_proj.bias"])[:, 0].numpy()
