"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product).

CPU restatement (numpy, float64 accumulation) of the reference's knowledge-distillation losses and
of their gradients with respect to the student scores.  PINNED: tests/golden/kd_loss.npz holds
losses and autograd gradients produced by the reference's own code (src/kd/losses.py, imported in
the build container by tests/golden/make_golden.py); tests/test_oracle_golden.py checks this
restatement against them.

Reference, scores ``s`` (student) and ``t`` (teacher) of shape [B, D]:
  MarginMSELoss   src/kd/losses.py:35-60    mse( s - max_j s , t/T - max_j t/T )  over all B*D
  ListwiseKDLoss  src/kd/losses.py:81-106   T^2 * KL( softmax(t/T) || softmax(s/T) ), batchmean
  ContrastiveLoss src/kd/losses.py:127-149  - mean_b log_softmax(s / tau)[b, 0]
  CombinedKDLoss  src/kd/losses.py:219-252  w_mm * MM + w_lk * LK + w_c * C  (tau = 0.05 fixed,
                  T annealed linearly from temperature_start to temperature_end, :200-217)
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def _log_softmax(x: np.ndarray) -> np.ndarray:
    m = x.max(axis=1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=1, keepdims=True))


def margin_mse(s: np.ndarray, t: np.ndarray, temperature: float) -> Tuple[float, np.ndarray]:
    """Loss and d loss / d s.  torch.max routes the gradient of the row maximum to ONE index (the
    first maximal one), which the gradient below reproduces."""
    s = s.astype(np.float64)
    ts = t.astype(np.float64) / temperature
    b, d = s.shape
    arg = s.argmax(axis=1)
    r = (s - s.max(axis=1, keepdims=True)) - (ts - ts.max(axis=1, keepdims=True))
    loss = float((r * r).mean())
    g = 2.0 * r / (b * d)
    g[np.arange(b), arg] -= 2.0 * r.sum(axis=1) / (b * d)
    return loss, g


def listwise_kd(s: np.ndarray, t: np.ndarray, temperature: float) -> Tuple[float, np.ndarray]:
    s = s.astype(np.float64)
    t = t.astype(np.float64)
    b = s.shape[0]
    ls = _log_softmax(s / temperature)
    lt = _log_softmax(t / temperature)
    pt = np.exp(lt)
    loss = float((pt * (lt - ls)).sum() / b * temperature**2)
    g = (np.exp(ls) - pt) * temperature / b
    return loss, g


def contrastive(s: np.ndarray, tau: float = 0.05) -> Tuple[float, np.ndarray]:
    s = s.astype(np.float64)
    b = s.shape[0]
    lp = _log_softmax(s / tau)
    loss = float(-lp[:, 0].mean())
    g = np.exp(lp)
    g[:, 0] -= 1.0
    return loss, g / (tau * b)


def combined(
    s: np.ndarray,
    t: np.ndarray,
    temperature: float = 4.0,
    weights: Tuple[float, float, float] = (0.6, 0.2, 0.2),
    tau: float = 0.05,
) -> Tuple[Dict[str, float], np.ndarray]:
    mm, gmm = margin_mse(s, t, temperature)
    lk, glk = listwise_kd(s, t, temperature)
    c, gc = contrastive(s, tau)
    w1, w2, w3 = weights
    out = {"loss": w1 * mm + w2 * lk + w3 * c, "margin_mse": mm, "listwise_kd": lk, "contrastive": c,
           "temperature": temperature}
    return out, w1 * gmm + w2 * glk + w3 * gc


def annealed_temperature(start: float, end: float, progress: float) -> float:
    """src/kd/losses.py:200-211."""
    return start + (end - start) * progress
