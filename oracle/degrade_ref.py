"""Oracle restatement of the reference's second-order degradation loop body and of one RealESRNet step
(reference: train_realesrnet.py:262-397; the same body is duplicated at train_realesrgan.py:342-457).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py -- never
by the product.  CPU fp32 torch, same op order as the reference.  Pinned against the reference's own `train()` executed
for one batch (tests/golden/gen_pipeline_golden.py -> tests/golden/pipeline_seed*.npz): every intermediate, the
quantised LR crop, the L1 loss and all 702 gradient norms (tests/test_oracle_pipeline_golden.py).

Randomness.  The reference mixes host draws (`random`, `np.random`) with device draws (torch).  Here the host
decisions arrive as a `plan` dict and the device draws through a `Draws` source: either recorded tensors (parity
tests replay the reference's own draws) or torch's global generator in the reference's call order.
"""
from __future__ import annotations

import random
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import imgproc_ref as I
from . import model_ref as M


class Draws:
    """Device-side random draws of the loop body in the reference's call order (imgproc.py:933-936 / 957-960: rand[B]
    for sigma or scale, rand[B] for the gray flag; :854 randn[h,w] only when a sample is gray, :858 randn[B,3,h,w];
    :895,:906 torch.poisson; train_realesrnet.py:307,355,361 uniform_[B] JPEG quality)."""

    def __init__(self, recorded: Optional[Iterable[Tuple[str, torch.Tensor]]] = None) -> None:
        self.rec = list(recorded) if recorded is not None else None
        self.pos = 0

    def _take(self, kind: str, shape) -> Optional[torch.Tensor]:
        if self.rec is None:
            return None
        k, t = self.rec[self.pos]
        self.pos += 1
        assert k == kind and tuple(t.shape) == tuple(shape), (self.pos - 1, k, kind, tuple(t.shape), tuple(shape))
        return t

    def rand(self, *shape):
        t = self._take("rand", shape)
        return torch.rand(*shape) if t is None else t

    def randn(self, *shape):
        t = self._take("randn", shape)
        return torch.randn(*shape) if t is None else t

    def poisson(self, lam):
        t = self._take("poisson", lam.shape)
        return torch.poisson(lam) if t is None else t

    def uniform(self, n, lo, hi):
        t = self._take("uniform", (n,))
        return torch.empty(n).uniform_(lo, hi) if t is None else t

    def exhausted(self) -> bool:
        return self.rec is None or self.pos == len(self.rec)


def sample_plan(hr_h: int, hr_w: int, crop: int, P: dict) -> Dict:
    """The loop body's HOST draws in the reference's order (train_realesrnet.py:275, 279-287, 291, 313, 317-325, 332,
    347, 351/368; imgproc.py:1913-1914).  Drawn up-front: the device ops between them never touch `random`/`np.random`."""
    def updown(prob, rng):
        kind = random.choices(["up", "down", "keep"], prob)[0]
        if kind == "up":
            return float(np.random.uniform(1, rng[1]))
        if kind == "down":
            return float(np.random.uniform(rng[0], 1))
        return 1.0
    p = {}
    p["blur1"] = bool(np.random.uniform() <= P["first_blur_probability"])
    p["resize1_scale"] = updown(P["resize_probability1"], P["resize_range1"])
    p["resize1_mode"] = random.choice(["area", "bilinear", "bicubic"])
    p["noise1_gaussian"] = bool(np.random.uniform() < P["gaussian_noise_probability1"])
    p["blur2"] = bool(np.random.uniform() < P["second_blur_probability"])
    p["resize2_scale"] = updown(P["resize_probability2"], P["resize_range2"])
    p["resize2_mode"] = random.choice(["area", "bilinear", "bicubic"])
    p["noise2_gaussian"] = bool(np.random.uniform() < P["gaussian_noise_probability2"])
    p["sinc_before_jpeg"] = bool(np.random.uniform() < 0.5)
    p["resize3_mode"] = random.choice(["area", "bilinear", "bicubic"])
    p["hr_top"] = random.randint(0, hr_h - crop)
    p["hr_left"] = random.randint(0, hr_w - crop)
    return p


def _noise(out, gaussian: bool, rng, poisson_rng, gray_prob, draws: Draws):
    """train_realesrnet.py:291-304 / 332-345 (clip=True, rounds=False)."""
    b = out.shape[0]
    if gaussian:                                                       # imgproc.py:1029-1057 -> 919-940 -> 829-863
        sigma = draws.rand(b) * (rng[1] - rng[0]) + rng[0]
        gray = (draws.rand(b) < gray_prob).float()
        fg = draws.randn(*out.shape[2:]) if gray.sum() > 0 else torch.zeros(out.shape[2:])
        fc = draws.randn(*out.shape)
        return I._finish(out + I.gaussian_noise_from_fields(out, sigma, gray, fg, fc), True, False)
    scale = draws.rand(b) * (poisson_rng[1] - poisson_rng[0]) + poisson_rng[0]   # imgproc.py:1060-1086 -> 943-964 -> 866-916
    gray = (draws.rand(b) < gray_prob).float()
    return I._finish(out + I.poisson_noise(out, scale, gray, sampler=draws.poisson), True, False)


def degrade_batch(hr: torch.Tensor, kernel1: torch.Tensor, kernel2: torch.Tensor, sinc_kernel: torch.Tensor, plan: Dict,
                  P: dict, upscale: int, crop: int, draws: Optional[Draws] = None, trace: Optional[dict] = None):
    """train_realesrnet.py:262-377 for one batch.  Returns (lr, hr_crop); `trace` receives every intermediate under the
    names the golden generator uses."""
    draws = draws or Draws()
    tr = trace if trace is not None else {}
    usm_k = I.usm_kernel(50, 0)
    out = tr["usm"] = I.usm_sharp(hr, usm_k, 0.5, 10)                                              # :268
    H, W = out.shape[2:]
    if plan["blur1"]:
        out = tr["blur1"] = I.filter2d(out, kernel1)                                               # :275-276
    out = tr["resize1"] = F.interpolate(out, scale_factor=plan["resize1_scale"], mode=plan["resize1_mode"])   # :279-288
    out = tr["noise1"] = _noise(out, plan["noise1_gaussian"], P["noise_range1"], P["poisson_scale_range1"],
                                P["gray_noise_probability1"], draws)                             # :291-304
    q = draws.uniform(out.shape[0], *P["jpeg_range1"])                                            # :307
    tr["q1"] = q.clone()
    out = tr["jpeg1"] = I.diff_jpeg(torch.clamp(out, 0, 1), q)                                    # :308-309
    if plan["blur2"]:
        out = tr["blur2"] = I.filter2d(out, kernel2)                                               # :313-314
    size2 = (int(H / upscale * plan["resize2_scale"]), int(W / upscale * plan["resize2_scale"]))  # :326-329
    out = tr["resize2"] = F.interpolate(out, size=size2, mode=plan["resize2_mode"])
    out = tr["noise2"] = _noise(out, plan["noise2_gaussian"], P["noise_range2"], P["poisson_scale_range2"],
                                P["gray_noise_probability2"], draws)                             # :332-345
    size3 = (H // upscale, W // upscale)
    if plan["sinc_before_jpeg"]:                                                                    # :347-358
        out = tr["resize3"] = F.interpolate(out, size=size3, mode=plan["resize3_mode"])
        out = tr["sinc"] = I.filter2d(out, sinc_kernel)
        q = draws.uniform(out.shape[0], *P["jpeg_range2"])
        tr["q2"] = q.clone()
        out = tr["jpeg2"] = I.diff_jpeg(torch.clamp(out, 0, 1), q)
    else:                                                                                           # :359-371
        q = draws.uniform(out.shape[0], *P["jpeg_range2"])
        tr["q2"] = q.clone()
        out = tr["jpeg2"] = I.diff_jpeg(torch.clamp(out, 0, 1), q)
        out = tr["resize3"] = F.interpolate(out, size=size3, mode=plan["resize3_mode"])
        out = tr["sinc"] = I.filter2d(out, sinc_kernel)
    lr_full = tr["lr_full"] = I.quantize(out)                                                      # :374
    return I.crop_pair(lr_full, hr, crop, upscale, plan["hr_top"], plan["hr_left"])               # :377; HR target is un-sharpened


def realesrnet_step(params: Dict[str, torch.Tensor], lr: torch.Tensor, hr: torch.Tensor, upscale: int = 4,
                    n_blocks: int = 23):
    """train_realesrnet.py:379-388 on the CPU (autocast is a no-op there): sr = G(lr); L1(sr, hr); backward.
    `params` must require grad; returns (loss, sr)."""
    sr = M.generator_forward(lr, params, upscale, n_blocks)
    loss = F.l1_loss(sr, hr)
    loss.backward()
    return loss.detach(), sr.detach()


def realesrgan_step(gparams: Dict[str, torch.Tensor], dstate: Dict[str, torch.Tensor], lr: torch.Tensor, hr: torch.Tensor,
                    pixel_weight: float = 1.0, adversarial_weight: float = 0.1, upscale: int = 4, n_blocks: int = 23):
    """train_realesrgan.py:459-516 on the CPU, gradients only (no optimiser): generator loss with the discriminator
    frozen -- pixel L1 on usm_sharpener(sr) + adversarial BCE(D(sr), 1), the VGG term being detached in the reference
    (:477-478) -- then BCE(D(hr), 1) and BCE(D(sr.detach()), 0) accumulated into the discriminator's gradients.
    `gparams` and the non-buffer entries of `dstate` must require grad; the spectral-norm u / v of `dstate` are updated in
    place by the three training-mode discriminator calls, as torch's hook does.  Returns the four loss values and sr."""
    usm_k = I.usm_kernel(50, 0)
    b, _, h, w = hr.shape
    real, fake = torch.ones(b, 1, h, w), torch.zeros(b, 1, h, w)                         # :460-461
    trainable = [v for k, v in dstate.items() if not (k.endswith("_u") or k.endswith("_v"))]
    for v in trainable:                                                                    # :465-466
        v.requires_grad_(False)
    sr = M.generator_forward(lr, gparams, upscale, n_blocks)                               # :474
    pixel = pixel_weight * F.l1_loss(I.usm_sharp(sr, usm_k, 0.5, 10), hr)                 # :475
    adv = adversarial_weight * F.binary_cross_entropy_with_logits(M.discriminator_forward(sr, dstate, True), real)   # :478
    (pixel + adv).backward()                                                               # :483
    for v in trainable:                                                                    # :491-492
        v.requires_grad_(True)
        v.grad = None                                                                      # :495
    d_hr = F.binary_cross_entropy_with_logits(M.discriminator_forward(hr, dstate, True), real)            # :499-500
    d_hr.backward()                                                                        # :503
    d_sr = F.binary_cross_entropy_with_logits(M.discriminator_forward(sr.detach().clone(), dstate, True), fake)   # :507-508
    d_sr.backward()                                                                        # :513
    return {"pixel_loss": pixel.detach(), "adversarial_loss": adv.detach(), "d_loss_hr": d_hr.detach(),
            "d_loss_sr": d_sr.detach()}, sr.detach()
