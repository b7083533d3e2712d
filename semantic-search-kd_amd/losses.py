"""Knowledge-distillation losses on the MI355X backend, with the reference's interface.

Mirror of ``src/kd/losses.py`` (``MarginMSELoss``, ``ListwiseKDLoss``, ``ContrastiveLoss``,
``CombinedKDLoss``: same constructor arguments, same ``forward`` signatures and return values, same
``update_temperature``), so the reference's training step (``src/kd/train.py:176-210``) and its
own tests (``tests/test_losses.py``) work against it unchanged.  Losses and the gradient with
respect to the student scores come from one HIP kernel pair (``csrc/kd_loss.hip``) wrapped in a
``torch.autograd.Function``; score matrices are ``[batch, n_docs]`` with ``n_docs <= 64``.
There is no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import _native

CONTRASTIVE_TEMPERATURE = 0.05  # src/kd/losses.py:190


def _check(scores: torch.Tensor, name: str) -> torch.Tensor:
    if not scores.is_cuda:
        raise RuntimeError(f"{name}: the MI355X backend has no CPU path; move the scores to the GPU")
    if scores.dim() != 2:
        raise ValueError(f"{name}: expected [batch, n_docs] scores, got {tuple(scores.shape)}")
    return scores


class _KDLossFunction(torch.autograd.Function):
    """forward -> (weighted total, [margin_mse, listwise_kd, contrastive]); backward: d total / d student."""

    @staticmethod
    def forward(ctx, student, teacher, temperature, tau, w_mm, w_lk, w_c):
        lib = _native.load()
        s = _check(student, "student_scores").detach().to(torch.float32).contiguous()
        t = _check(teacher, "teacher_scores").detach().to(torch.float32).contiguous()
        if s.shape != t.shape:
            raise ValueError(f"student {tuple(s.shape)} and teacher {tuple(t.shape)} score shapes differ")
        b, d = s.shape
        losses = torch.empty(4, dtype=torch.float32, device=s.device)
        grad = torch.empty_like(s)
        rows = torch.empty(3 * b, dtype=torch.float32, device=s.device)
        with torch.cuda.device(s.device):
            stream = int(torch.cuda.current_stream(s.device).cuda_stream)
            _native.check(
                lib.sskd_kd_loss(
                    s.data_ptr(), t.data_ptr(), b, d, float(temperature), float(tau),
                    float(w_mm), float(w_lk), float(w_c),
                    losses.data_ptr(), grad.data_ptr(), rows.data_ptr(), stream,
                )
            )
        ctx.save_for_backward(grad)
        ctx.in_dtype = student.dtype
        components = losses[1:]
        ctx.mark_non_differentiable(components)
        return losses[0], components

    @staticmethod
    def backward(ctx, grad_total, _grad_components):
        (grad,) = ctx.saved_tensors
        return (grad * grad_total).to(ctx.in_dtype), None, None, None, None, None, None


class MarginMSELoss(nn.Module):
    """src/kd/losses.py:16-60: MSE between max-subtracted student and (temperature-softened) teacher scores."""

    def __init__(self, temperature: float = 1.0):
        super().__init__()
        self.temperature = temperature

    def forward(self, student_scores: torch.Tensor, teacher_scores: torch.Tensor) -> torch.Tensor:
        total, _ = _KDLossFunction.apply(student_scores, teacher_scores, self.temperature,
                                         CONTRASTIVE_TEMPERATURE, 1.0, 0.0, 0.0)
        return total


class ListwiseKDLoss(nn.Module):
    """src/kd/losses.py:63-106: T^2 * KL(softmax(teacher / T) || softmax(student / T)), batch mean."""

    def __init__(self, temperature: float = 1.0):
        super().__init__()
        self.temperature = temperature

    def forward(self, student_scores: torch.Tensor, teacher_scores: torch.Tensor) -> torch.Tensor:
        total, _ = _KDLossFunction.apply(student_scores, teacher_scores, self.temperature,
                                         CONTRASTIVE_TEMPERATURE, 0.0, 1.0, 0.0)
        return total


class ContrastiveLoss(nn.Module):
    """src/kd/losses.py:109-149: InfoNCE with the first document as the positive."""

    def __init__(self, temperature: float = 0.05):
        super().__init__()
        self.temperature = temperature

    def forward(self, student_scores: torch.Tensor) -> torch.Tensor:
        total, _ = _KDLossFunction.apply(student_scores, student_scores.detach(), 1.0,
                                         self.temperature, 0.0, 0.0, 1.0)
        return total


class CombinedKDLoss(nn.Module):
    """src/kd/losses.py:152-252: weighted sum of the three losses with a linearly annealed temperature."""

    def __init__(
        self,
        margin_mse_weight: float = 0.6,
        listwise_kd_weight: float = 0.2,
        contrastive_weight: float = 0.2,
        temperature_start: float = 4.0,
        temperature_end: float = 2.0,
    ):
        super().__init__()
        self.margin_mse_weight = margin_mse_weight
        self.listwise_kd_weight = listwise_kd_weight
        self.contrastive_weight = contrastive_weight
        self.temperature_start = temperature_start
        self.temperature_end = temperature_end
        self.current_temperature = temperature_start
        self.component_tensors = False   # True: the three component losses come back as 0-dim device tensors (no host sync)
        self.margin_mse_loss = MarginMSELoss(temperature=temperature_start)
        self.listwise_kd_loss = ListwiseKDLoss(temperature=temperature_start)
        self.contrastive_loss = ContrastiveLoss(temperature=CONTRASTIVE_TEMPERATURE)

    def update_temperature(self, progress: float) -> None:
        self.current_temperature = self.temperature_start + (self.temperature_end - self.temperature_start) * progress
        self.margin_mse_loss.temperature = self.current_temperature
        self.listwise_kd_loss.temperature = self.current_temperature

    def forward(self, student_scores: torch.Tensor, teacher_scores: torch.Tensor) -> Dict[str, object]:
        if self.margin_mse_loss.temperature != self.listwise_kd_loss.temperature:
            # the two component temperatures were set apart by hand: evaluate them separately
            mm = self.margin_mse_loss(student_scores, teacher_scores)
            lk = self.listwise_kd_loss(student_scores, teacher_scores)
            c = self.contrastive_loss(student_scores)
            total = self.margin_mse_weight * mm + self.listwise_kd_weight * lk + self.contrastive_weight * c
            return {"loss": total, "margin_mse": mm.item(), "listwise_kd": lk.item(), "contrastive": c.item(),
                    "temperature": self.current_temperature}
        total, comps = _KDLossFunction.apply(
            student_scores, teacher_scores, self.margin_mse_loss.temperature, self.contrastive_loss.temperature,
            self.margin_mse_weight, self.listwise_kd_weight, self.contrastive_weight,
        )
        if self.component_tensors or torch.cuda.is_current_stream_capturing():
            # no host read: inside a HIP-graph capture (training.GraphedStep) a device-to-host copy is not permitted, and
            # in an eager step it is a synchronisation point between the forward and the backward pass
            mm, lk, c = comps[0], comps[1], comps[2]
        else:
            mm, lk, c = (float(v) for v in comps.tolist())  # the reference returns .item() floats here too
        return {
            "loss": total,
            "margin_mse": mm,
            "listwise_kd": lk,
            "contrastive": c,
            "temperature": self.current_temperature,
        }
