"""The scalar losses of the train steps on one launch each (csrc/loss.hip).

Reference call sites: `nn.L1Loss` (train_realesrnet.py:385, train_realesrgan.py:475) and `nn.BCEWithLogitsLoss` against
`torch.full(..., 1.0)` / `torch.full(..., 0.0)` label tensors (train_realesrgan.py:460-461,478,500,509).  `l1_loss` /
`bce_with_logits_const` compute `weight * criterion(...)` -- the value AND the unit gradient in the same launch, the label never
materialised -- when `criterion` is the stock module in its default configuration (what the reference constructs,
train_realesrgan.py:178-180); any other criterion (a subclass, a reduction other than "mean", class weights) is simply called.
As stock ATen ops these losses were ~56 launches / 1.8 ms of a 28 ms RealESRGAN step at 16 x 256^2.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib

_scratch: Dict[tuple, torch.Tensor] = {}


def _scratch_for(device: torch.device, slot: str) -> torch.Tensor:
    """Zero-filled once, then owned by the kernels (arrival counter + partials); one buffer per call site (`slot`) and stream,
    so losses of different streams never share a counter."""
    key = (device.type, device.index, slot, torch.cuda.current_stream(device).cuda_stream)
    t = _scratch.get(key)
    if t is None:
        t = torch.zeros(int(_lib.lib().resr_loss_scratch_bytes()) // 4, dtype=torch.float32, device=device)
        _scratch[key] = t
    return t


def _plain(t: torch.Tensor) -> bool:
    if os.environ.get("RESR_UNFUSED_LOSSES") == "1":      # A/B knob (tools/prof_gan.sh): the stock ATen criteria
        return False
    return t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a: torch.Tensor, b: torch.Tensor, weight: float, slot: str):
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        need = a.requires_grad or b.requires_grad
        grad = torch.empty_like(a) if need else None
        _lib.check(_lib.lib().resr_l1_mean(_lib.ptr(a), _lib.ptr(b), a.numel(), float(weight), _lib.ptr(loss), _lib.ptr(grad),
                                           _lib.ptr(_scratch_for(a.device, slot)), _lib.stream_ptr(a)), "resr_l1_mean")
        ctx.grad = grad
        ctx.needs = (a.requires_grad, b.requires_grad)
        return loss

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        ga = ctx.grad * g if ctx.needs[0] else None          # g: the 0-d upstream gradient (loss scale included), on the device
        gb = (ctx.grad * (-g)) if ctx.needs[1] else None
        return ga, gb, None, None


class _BCELogitsConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, label: float, weight: float, slot: str):
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x) if x.requires_grad else None
        _lib.check(_lib.lib().resr_bce_logits_const(_lib.ptr(x), x.numel(), float(label), float(weight), _lib.ptr(loss), _lib.ptr(grad),
                                                    _lib.ptr(_scratch_for(x.device, slot)), _lib.stream_ptr(x)), "resr_bce_logits_const")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None


def l1_loss(criterion: nn.Module, a: torch.Tensor, b: torch.Tensor, weight: float = 1.0, slot: str = "l1") -> torch.Tensor:
    """`weight * criterion(a, b)`; one launch when `criterion` is a stock `nn.L1Loss()` (reduction "mean")."""
    if (type(criterion) is nn.L1Loss and criterion.reduction == "mean" and a.shape == b.shape and _plain(a) and _plain(b)
            and a.numel() > 0):
        return _L1Mean.apply(a, b, float(weight), slot)
    out = criterion(a, b)
    return out if weight == 1.0 else weight * out


def bce_with_logits_const(criterion: nn.Module, logits: torch.Tensor, label: float, weight: float = 1.0,
                          slot: str = "bce") -> torch.Tensor:
    """`weight * criterion(logits, torch.full_like(logits, label))` (train_realesrgan.py:460-461,478,500,509); one launch, the
    label tensor never built, when `criterion` is a stock `nn.BCEWithLogitsLoss()` (mean, no class / positive weights)."""
    if (type(criterion) is nn.BCEWithLogitsLoss and criterion.reduction == "mean" and criterion.weight is None
            and criterion.pos_weight is None and _plain(logits) and logits.numel() > 0):
        return _BCELogitsConst.apply(logits, float(label), float(weight), slot)
    out = criterion(logits, torch.full_like(logits, float(label)))
    return out if weight == 1.0 else weight * out
