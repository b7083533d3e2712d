"""Encoder oracle (CPU, fp32) — TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates what the reference's ``StudentModel.encode`` executes inside its third-party
engines (sentence-transformers ^2.2.2 -> transformers ^4.35 ``BertModel``; pyproject.toml:11-15;
pipeline Transformer -> Pooling(mean) -> Normalize, tests/test_model_validation.py:80-89,256-262):

    x   = LN(word[id] + pos[t] + type[0])                                  eps 1e-12
    per layer (post-LN BERT):
      q,k,v = x Wq^T + bq, ...;  heads of 32
      a   = softmax(q k^T / sqrt(32) + (1 - mask) * -inf) v
      x   = LN(x + a Wo^T + bo)
      x   = LN(x + gelu_erf(x W1^T + b1) W2^T + b2)
    e   = sum_t m_t x_t / clamp(sum_t m_t, 1e-9);  e / max(||e||_2, 1e-12)

Pinned: ``tests/golden/make_golden.py`` checks this restatement against
``transformers.BertModel`` (built from an in-memory ``BertConfig``, no download) on the
synthetic weights and commits the resulting vectors under ``tests/golden/``.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch


def _ln(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def _gelu_erf(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / np.sqrt(2.0)))


def bert_hidden_states(
    sd: Dict[str, np.ndarray],
    input_ids: np.ndarray,
    attention_mask: np.ndarray,
    num_layers: int,
    num_heads: int = 12,
    eps: float = 1e-12,
    return_all: bool = False,
    dtype=torch.float32,
):
    """Final (or per-layer) hidden states ``[B, S, H]`` of a post-LN BERT encoder."""
    t = {k: torch.from_numpy(np.asarray(v, np.float32)).to(dtype) for k, v in sd.items()}
    outs = bert_hidden_states_torch(t, input_ids, attention_mask, num_layers, num_heads, eps, dtype)
    if return_all:
        return [o.float().numpy() for o in outs]
    return outs[-1].float().numpy()


def bert_hidden_states_torch(t, input_ids, attention_mask, num_layers, num_heads=12, eps=1e-12, dtype=torch.float32,
                             pos_offset: int = 0):
    """The same forward on a dict of torch tensors (which may require grad: torch autograd over this
    restatement is the gradient oracle of the training path); returns the list of per-layer states."""
    ids = torch.from_numpy(np.asarray(input_ids)).long()
    mask = torch.from_numpy(np.asarray(attention_mask)).to(dtype)
    B, S = ids.shape
    x = (
        t["embeddings.word_embeddings.weight"][ids]
        + t["embeddings.position_embeddings.weight"][pos_offset : pos_offset + S][None]
        + t["embeddings.token_type_embeddings.weight"][0][None, None]
    )
    x = _ln(x, t["embeddings.LayerNorm.weight"], t["embeddings.LayerNorm.bias"], eps)
    H = x.shape[-1]
    dh = H // num_heads
    bias = (1.0 - mask)[:, None, None, :] * torch.finfo(dtype).min
    outs: List[torch.Tensor] = [x]
    for i in range(num_layers):
        p = f"encoder.layer.{i}."

        def lin(name, inp):
            return inp @ t[p + name + ".weight"].T + t[p + name + ".bias"]

        def heads(y):
            return y.view(B, S, num_heads, dh).transpose(1, 2)

        q, k, v = heads(lin("attention.self.query", x)), heads(lin("attention.self.key", x)), heads(
            lin("attention.self.value", x)
        )
        sc = q @ k.transpose(-1, -2) / np.sqrt(dh) + bias
        a = torch.softmax(sc, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, S, H)
        x = _ln(
            x + lin("attention.output.dense", a),
            t[p + "attention.output.LayerNorm.weight"],
            t[p + "attention.output.LayerNorm.bias"],
            eps,
        )
        hmid = _gelu_erf(lin("intermediate.dense", x))
        x = _ln(x + lin("output.dense", hmid), t[p + "output.LayerNorm.weight"], t[p + "output.LayerNorm.bias"], eps)
        outs.append(x)
    return outs


def embeddings_torch(t, input_ids, attention_mask, num_layers, num_heads=12, eps=1e-12, normalize=True):
    """Differentiable masked mean-pool (+ L2 normalise) of ``bert_hidden_states_torch``."""
    h = bert_hidden_states_torch(t, input_ids, attention_mask, num_layers, num_heads, eps)[-1]
    m = torch.from_numpy(np.asarray(attention_mask)).to(h.dtype)[..., None]
    e = (h * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)
    if normalize:
        e = e / torch.clamp(e.norm(dim=1, keepdim=True), min=1e-12)
    return e


def mean_pool_normalize(hidden: np.ndarray, attention_mask: np.ndarray, normalize: bool = True) -> np.ndarray:
    """sentence-transformers ``Pooling(mean)`` + ``Normalize`` in numpy (float64 accumulation)."""
    h = np.asarray(hidden, np.float64)
    m = np.asarray(attention_mask, np.float64)[..., None]
    e = (h * m).sum(1) / np.clip(m.sum(1), 1e-9, None)
    if normalize:
        e = e / np.clip(np.linalg.norm(e, axis=1, keepdims=True), 1e-12, None)
    return e.astype(np.float32)


def encode_token_ids(
    sd: Dict[str, np.ndarray],
    input_ids: np.ndarray,
    attention_mask: Optional[np.ndarray],
    num_layers: int,
    num_heads: int = 12,
    eps: float = 1e-12,
    normalize: bool = True,
) -> np.ndarray:
    if attention_mask is None:
        attention_mask = np.ones_like(input_ids)
    hs = bert_hidden_states(sd, input_ids, attention_mask, num_layers, num_heads, eps)
    return mean_pool_normalize(hs, attention_mask, normalize)


def synthetic_token_ids(B: int, S: int, seed: int, vocab: int = 30522, lengths=None):
    """BASELINE.md §4 token recipe: ids uniform in [999, vocab), [CLS]=101 first, [SEP]=102 last
    real token, [PAD]=0 after; ``lengths`` (per row) gives ragged batches."""
    g = np.random.Generator(np.random.PCG64(seed))
    lo = min(999, vocab - 1)
    ids = g.integers(lo, vocab, size=(B, S), dtype=np.int64).astype(np.int32)
    mask = np.ones((B, S), np.int32)
    if lengths is None:
        lengths = [S] * B
    for b, n in enumerate(lengths):
        n = min(max(2, int(n)), S)
        ids[b, 0] = 101
        ids[b, n - 1] = 102
        ids[b, n:] = 0
        mask[b, n:] = 0
    return ids, mask
