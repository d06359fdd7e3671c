"""Degradation ops behind the reference's `imgproc.py` surface (same names, argument order, return
shapes; reference imgproc.py:29-38), executed by the HIP kernels of csrc/degrade.hip through the C-ABI.
Inputs must live on the MI355X; there is no CPU path.

Host-side blur / sinc kernel synthesis (reference imgproc.py:170-603, dataset.py:82-143) is float64
numpy, as in the reference's DataLoader workers: 3 kernels of 21x21 per sample.
"""
from __future__ import annotations

import math
import random
from typing import Any, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as _dist
from torch import nn

from . import _lib

__all__ = ["random_add_gaussian_noise_torch", "random_add_poisson_noise_torch", "random_mixed_kernels",
           "generate_sinc_kernel", "image_to_tensor", "tensor_to_image", "random_crop", "filter2d_torch",
           "interpolate", "filter2d_u8", "interpolate_u8", "jpeg_u8", "quantize_kernel_q14", "resize_tap_tables", "DiffJPEG", "USMSharp", "image_resize", "center_crop", "random_rotate",
           "random_horizontally_flip", "random_vertically_flip", "rgb2ycbcr_torch", "read_image_rgb"]

_MODES = {"area": 0, "bilinear": 1, "bicubic": 2}
_seed_counter = [0x5EED]


def _next_seed() -> int:
    """Philox seed of the next device-side noise draw: torch's seed, the data-parallel rank (replicas share
    torch.manual_seed, config.py:64-66, but must not draw the same noise fields) and a per-process counter."""
    _seed_counter[0] += 1
    rank = _dist.get_rank() if _dist.is_available() and _dist.is_initialized() else 0
    return ((torch.initial_seed() + 0x632BE59BD9B4E019 * rank) * 0x9E3779B97F4A7C15 + _seed_counter[0]) & 0xFFFFFFFFFFFFFFFF


def _img(x: torch.Tensor, what: str) -> torch.Tensor:
    _lib.require_cuda(x, what)
    if x.dim() != 4:
        raise ValueError(f"{what}: expected [N,C,H,W]")
    return x.float().contiguous()


# ---- filters -------------------------------------------------------------------------------------------
def filter2d_torch(image: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """cv2.filter2D equivalent (reference imgproc.py:1089-1121): reflect padding, correlation;
    kernel [1,k,k] shared by the batch or [B,k,k] per sample."""
    x = _img(image, "filter2d_torch")
    k = kernel.size(-1)
    if k % 2 != 1:
        raise ValueError("Wrong kernel size.")
    b, c, h, w = x.shape
    kk = kernel.to(device=x.device, dtype=torch.float32).contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().resr_filter2d(_lib.ptr(x), _lib.ptr(out), _lib.ptr(kk), b, c, h, w, k, k,
                                        0 if kernel.size(0) == 1 else 1, _lib.stream_ptr(x)), "resr_filter2d")
    return out


def _gaussian_1d(ksize: int, sigma: float) -> np.ndarray:
    # cv2.getGaussianKernel, general branch (reference imgproc.py:1522)
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


class USMSharp(nn.Module):
    """Unsharp masking (reference imgproc.py:1514-1537).  The [1,k,k] `kernel` buffer is kept for
    state_dict compatibility; the blur itself runs as the two separable k-tap passes of its factor."""

    def __init__(self, radius: int, sigma: int) -> None:
        super().__init__()
        if radius % 2 == 0:
            radius += 1
        self.radius = radius
        g = _gaussian_1d(radius, sigma)
        self.register_buffer("kernel", torch.from_numpy(np.outer(g, g).astype(np.float32)).unsqueeze_(0))
        self.register_buffer("k1d", torch.from_numpy(g.astype(np.float32)), persistent=False)

    def forward(self, x, weight: float, threshold: int) -> torch.Tensor:
        # differentiable wrt x (the soft mask is piecewise constant), as the reference's GAN step requires
        return _USMFn.apply(x, self.k1d, self.radius, float(weight), float(threshold))


class _USMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k1d, radius, weight, threshold):
        xi = _img(x, "USMSharp")
        b, c, h, w = xi.shape
        out = torch.empty_like(xi)
        tmp = torch.empty(3 * xi.numel(), dtype=torch.float32, device=xi.device)
        # (the soft mask is stored only for a backward pass: the degradation path sharpens without a graph)
        fn = _lib.lib().resr_usm_sharp if x.requires_grad else _lib.lib().resr_usm_sharp_forward_only
        _lib.check(fn(_lib.ptr(xi), _lib.ptr(out), _lib.ptr(tmp), _lib.ptr(k1d), radius, weight,
                      threshold, b, c, h, w, _lib.stream_ptr(x)), "resr_usm_sharp")
        if x.requires_grad:
            ctx.save_for_backward(xi, tmp, k1d)
            ctx.radius, ctx.weight = radius, weight
        return out

    @staticmethod
    def backward(ctx, g):
        xi, tmp, k1d = ctx.saved_tensors
        b, c, h, w = xi.shape
        gi = g.contiguous().float()
        gx = torch.empty_like(xi)
        tmp2 = torch.empty(2 * xi.numel(), dtype=torch.float32, device=xi.device)
        _lib.check(_lib.lib().resr_usm_sharp_bwd(_lib.ptr(xi), _lib.ptr(tmp), _lib.ptr(gi), _lib.ptr(gx), _lib.ptr(tmp2),
                                                 _lib.ptr(k1d), ctx.radius, ctx.weight, b, c, h, w, _lib.stream_ptr(g)),
                   "resr_usm_sharp_bwd")
        return gx, None, None, None, None


def interpolate(image: torch.Tensor, size=None, scale_factor=None, mode: str = "bilinear") -> torch.Tensor:
    """F.interpolate(mode in area|bilinear|bicubic, align_corners=False) as used at reference
    train_realesrnet.py:288 (scale_factor=) and :326-329, :349-351, :366-368 (size=)."""
    x = _img(image, "interpolate")
    b, c, h, w = x.shape
    if (size is None) == (scale_factor is None):
        raise ValueError("give exactly one of size / scale_factor")
    if size is not None:
        oh, ow = (size, size) if isinstance(size, int) else size
        sh = sw = 0.0
    else:
        sh = sw = float(scale_factor)
        oh, ow = int(math.floor(float(h) * sh)), int(math.floor(float(w) * sw))
    out = torch.empty((b, c, oh, ow), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().resr_resize(_lib.ptr(x), _lib.ptr(out), b, c, h, w, oh, ow, _MODES[mode], sh, sw,
                                      _lib.stream_ptr(x)), "resr_resize")
    return out


# ---- integer mode of blur / resize (north_star "bit-exact in integer mode") ----------------
# uint8 images, fixed-point taps, integer accumulation on the device (csrc/degrade_int.hip): bit-identical to the CPU
# restatement whatever the summation order.  The float path above stays the training path (it is what the reference runs).
Q_KERNEL, Q_RESIZE = 14, 11


def quantize_kernel_q14(kernel) -> np.ndarray:
    """Blur kernel(s) [..., kh, kw] (float) -> Q14 int32 taps whose sum per kernel is EXACTLY round(sum * 2^14): rounded to
    nearest-even, then the rounding residue is put on the largest tap (so a normalised kernel keeps unit DC gain)."""
    k = np.asarray(kernel.detach().cpu().numpy() if torch.is_tensor(kernel) else kernel, dtype=np.float64)
    q = np.rint(k * (1 << Q_KERNEL)).astype(np.int64)
    flat = q.reshape(-1, k.shape[-2] * k.shape[-1])
    want = np.rint(k.reshape(flat.shape).sum(axis=1) * (1 << Q_KERNEL)).astype(np.int64)
    peak = np.abs(flat).argmax(axis=1)
    flat[np.arange(flat.shape[0]), peak] += want - flat.sum(axis=1)
    return flat.reshape(q.shape).astype(np.int32)


def filter2d_u8(image: torch.Tensor, kernel) -> torch.Tensor:
    """Integer-mode `filter2d_torch` (reference imgproc.py:1089-1121): uint8 [N,C,H,W] in, uint8 out; `kernel` float
    [1|N, k, k] (quantised here) or already-quantised int32 taps."""
    _lib.require_cuda(image, "filter2d_u8")
    if image.dtype != torch.uint8 or image.dim() != 4:
        raise ValueError("filter2d_u8: expected a uint8 [N,C,H,W] image")
    x = image.contiguous()
    b, c, h, w = x.shape
    taps = kernel if (torch.is_tensor(kernel) and kernel.dtype == torch.int32) else torch.from_numpy(quantize_kernel_q14(kernel))
    if taps.dim() != 3 or taps.shape[0] not in (1, b):
        raise ValueError("filter2d_u8: kernel must be [1 or N, kh, kw]")
    taps = taps.to(x.device).contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().resr_filter2d_u8(_lib.ptr(x), _lib.ptr(out), _lib.ptr(taps), b, c, h, w, taps.shape[1], taps.shape[2],
                                           0 if taps.shape[0] == 1 else 1, _lib.stream_ptr(x)), "resr_filter2d_u8")
    return out


def resize_tap_tables(in_size: int, out_size: int, scale: Optional[float], mode: str):
    """Per-axis tap tables of F.interpolate(align_corners=False, antialias=False) in fixed point: (idx [out, taps] int32
    clamped source indices, w [out, taps] int32 Q11 weights summing to 2^11).  Coordinates follow ATen
    (area_pixel_compute_source_index): src = (dst + 0.5) * s - 0.5 with s = 1/scale_factor when the caller gave
    scale_factor= (train_realesrnet.py:288), in/out for size=; bilinear clamps src at 0, bicubic (A = -0.75) clamps the
    tap indices instead.  float64 on the host, one table per call."""
    s = (1.0 / scale) if scale else in_size / out_size
    src = (np.arange(out_size, dtype=np.float64) + 0.5) * s - 0.5
    if mode == "bilinear":
        src = np.maximum(src, 0.0)
        i0 = np.minimum(np.floor(src).astype(np.int64), in_size - 1)
        lam = src - i0
        idx = np.stack([i0, np.minimum(i0 + 1, in_size - 1)], axis=1)
        wts = np.stack([1.0 - lam, lam], axis=1)
    elif mode == "bicubic":
        A = -0.75
        i0 = np.floor(src).astype(np.int64)
        t = src - i0

        def c1(x):
            return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

        def c2(x):
            return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
        wts = np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], axis=1)
        idx = np.clip(np.stack([i0 - 1, i0, i0 + 1, i0 + 2], axis=1), 0, in_size - 1)
    else:
        raise ValueError("tap tables exist for bilinear / bicubic")
    q = np.rint(wts * (1 << Q_RESIZE)).astype(np.int64)
    peak = np.abs(q).argmax(axis=1)
    q[np.arange(out_size), peak] += (1 << Q_RESIZE) - q.sum(axis=1)
    return idx.astype(np.int32), q.astype(np.int32)


def interpolate_u8(image: torch.Tensor, size=None, scale_factor=None, mode: str = "bilinear") -> torch.Tensor:
    """Integer-mode `interpolate` (same call forms as above): uint8 in, uint8 out."""
    _lib.require_cuda(image, "interpolate_u8")
    if image.dtype != torch.uint8 or image.dim() != 4:
        raise ValueError("interpolate_u8: expected a uint8 [N,C,H,W] image")
    x = image.contiguous()
    b, c, h, w = x.shape
    if (size is None) == (scale_factor is None):
        raise ValueError("give exactly one of size / scale_factor")
    if size is not None:
        oh, ow = (size, size) if isinstance(size, int) else size
        sc = None
    else:
        sc = float(scale_factor)
        oh, ow = int(math.floor(float(h) * sc)), int(math.floor(float(w) * sc))
    out = torch.empty((b, c, oh, ow), dtype=torch.uint8, device=x.device)
    tabs = [None] * 4
    if mode != "area":
        iy, wy = resize_tap_tables(h, oh, sc, mode)
        ix, wx = resize_tap_tables(w, ow, sc, mode)
        tabs = [torch.from_numpy(t).to(x.device) for t in (iy, wy, ix, wx)]
    _lib.check(_lib.lib().resr_resize_u8(_lib.ptr(x), _lib.ptr(out), b, c, h, w, oh, ow, _MODES[mode], *[_lib.ptr(t) for t in tabs],
                                         _lib.stream_ptr(x)), "resr_resize_u8")
    return out


# ---- noise ---------------------------------------------------------------------------------------------
def _finish_mode(clip: bool, rounds: bool) -> int:
    return (1 if clip else 0) | (2 if rounds else 0)


def add_gaussian_noise_fields(image, sigma, gray, field_gray, field_color, clip=True, rounds=False):
    """Gaussian noise with every random draw given (parity tests inject the reference's draws)."""
    x = _img(image, "gaussian noise")
    b, c, h, w = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.lib().resr_noise_gaussian(_lib.ptr(x), _lib.ptr(out), _lib.ptr(sigma.float().contiguous()),
                                              _lib.ptr(gray.float().contiguous()),
                                              _lib.ptr(field_gray.contiguous()) if field_gray is not None else None,
                                              _lib.ptr(field_color.contiguous()), b, c, h, w, _finish_mode(clip, rounds),
                                              _lib.stream_ptr(x)), "resr_noise_gaussian")
    return out


def random_add_gaussian_noise_torch(image: torch.Tensor, sigma_range: tuple = (0, 1.0), gray_prob: int = 0,
                                    clip: bool = True, rounds: bool = False) -> torch.Tensor:
    """Reference imgproc.py:1029-1057.  Per-sample sigma ~ U(range), gray flag ~ Bernoulli(gray_prob); the
    gray noise is ONE h x w field shared by the whole batch (reference quirk, imgproc.py:854-855).  All draws
    stay on the device (Philox); nothing is read back."""
    x = _img(image, "random_add_gaussian_noise_torch")
    b, c, h, w = x.shape
    sigma = torch.rand(b, device=x.device) * (sigma_range[1] - sigma_range[0]) + sigma_range[0]
    gray = (torch.rand(b, device=x.device) < gray_prob).float()
    fields = torch.empty(h * w + x.numel(), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().resr_randn_fill(_lib.ptr(fields), fields.numel(), _next_seed(), 0, _lib.stream_ptr(fields)), "resr_randn_fill")
    return add_gaussian_noise_fields(x, sigma, gray, fields[:h * w], fields[h * w:], clip, rounds)


def add_poisson_noise(image, scale, gray, seed: int, clip=True, rounds=False, return_vals=False):
    x = _img(image, "poisson noise")
    b, c, h, w = x.shape
    out = torch.empty_like(x)
    ws = torch.empty(_lib.lib().resr_noise_poisson_workspace_bytes(b), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().resr_noise_poisson(_lib.ptr(x), _lib.ptr(out), _lib.ptr(scale.float().contiguous()),
                                             _lib.ptr(gray.float().contiguous()), seed, _lib.ptr(ws), b, c, h, w,
                                             _finish_mode(clip, rounds), _lib.stream_ptr(x)), "resr_noise_poisson")
    if return_vals:
        return out, ws[b * 2048:].view(torch.float32).view(b, 2)
    return out


def random_add_poisson_noise_torch(image: torch.Tensor, scale_range: tuple = (0, 1.0), gray_prob: int = 0,
                                   clip: bool = True, rounds: bool = False) -> torch.Tensor:
    """Reference imgproc.py:1060-1086; the per-sample `len(torch.unique(...))` host loop (imgproc.py:892,903)
    is a 256-bin presence bitmap on the device -- no synchronisation."""
    x = _img(image, "random_add_poisson_noise_torch")
    b = x.shape[0]
    scale = torch.rand(b, device=x.device) * (scale_range[1] - scale_range[0]) + scale_range[0]
    gray = (torch.rand(b, device=x.device) < gray_prob).float()
    return add_poisson_noise(x, scale, gray, _next_seed(), clip, rounds)


# ---- JPEG ----------------------------------------------------------------------------------------------
class DiffJPEG(nn.Module):
    """DiffJPEG (reference imgproc.py:1462-1494), non-differentiable rounding only (the reference's train
    loops construct DiffJPEG(False), train_realesrnet.py:231).  `quality` may be a number or a [B] tensor;
    like the reference, a tensor argument is overwritten in place with the quantisation factors."""

    def __init__(self, differentiable: bool) -> None:
        super().__init__()
        if differentiable:
            raise NotImplementedError("the MI355X path implements DiffJPEG(differentiable=False), the only form the "
                                      "reference's training loops use (train_realesrnet.py:231, train_realesrgan.py)")

    def forward(self, x: torch.Tensor, quality, return_coeffs: bool = False, clamp_input: bool = False):
        xi = _img(x, "DiffJPEG")
        b, c, h, w = xi.shape
        if c != 3:
            raise ValueError("DiffJPEG expects RGB input")
        if isinstance(quality, (int, float)):
            q = torch.full((b,), float(quality), dtype=torch.float32, device=xi.device)
        else:
            q = quality.to(device=xi.device, dtype=torch.float32).contiguous().clone()
        out = torch.empty_like(xi)
        coeffs = None
        if return_coeffs:
            mb = ((h + 15) // 16) * ((w + 15) // 16)
            coeffs = torch.empty((b, mb * 6, 64), dtype=torch.float32, device=xi.device)
        _lib.check(_lib.lib().resr_jpeg(_lib.ptr(xi), _lib.ptr(out), _lib.ptr(q), _lib.ptr(coeffs), b, h, w,
                                        1 if clamp_input else 0, _lib.stream_ptr(xi)), "resr_jpeg")
        if isinstance(quality, torch.Tensor):       # reference quirk: factor[i] = f(quality[i]) written into the caller's tensor
            quality.copy_(torch.where(q < 50, (5000.0 / q) / 100.0, (200.0 - q * 2) / 100.0))
        return (out, coeffs) if return_coeffs else out


def jpeg_u8(image: torch.Tensor, quality, return_coeffs: bool = False):
    """Integer-mode `DiffJPEG(False).forward` (reference imgproc.py:1462-1494): uint8 RGB [N,3,H,W] in, uint8 out, fixed-point
    constants and integer accumulation throughout (include/resr.h: resr_jpeg_u8).  `quality`: a number or a [N] tensor, which
    -- as in the reference -- is overwritten in place with the quantisation factors.  With `return_coeffs` also the quantised
    coefficients, int32 [N, blocks, 64] in DiffJPEG.forward(return_coeffs=True)'s block order."""
    _lib.require_cuda(image, "jpeg_u8")
    if image.dtype != torch.uint8 or image.dim() != 4 or image.shape[1] != 3:
        raise ValueError("jpeg_u8: expected a uint8 [N,3,H,W] image")
    x = image.contiguous()
    b, _, h, w = x.shape
    if isinstance(quality, (int, float)):
        q = torch.full((b,), float(quality), dtype=torch.float32, device=x.device)
    else:
        q = quality.to(device=x.device, dtype=torch.float32).contiguous().clone()
    if q.numel() != b:
        raise ValueError("jpeg_u8: quality must be a number or one value per image")
    out = torch.empty_like(x)
    coeffs = None
    if return_coeffs:
        mb = ((h + 15) // 16) * ((w + 15) // 16)
        coeffs = torch.empty((b, mb * 6, 64), dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().resr_jpeg_u8(_lib.ptr(x), _lib.ptr(out), _lib.ptr(q), _lib.ptr(coeffs), b, h, w, _lib.stream_ptr(x)),
               "resr_jpeg_u8")
    if isinstance(quality, torch.Tensor):
        quality.copy_(torch.where(q < 50, (5000.0 / q) / 100.0, (200.0 - q * 2) / 100.0))
    return (out, coeffs) if return_coeffs else out


# ---- crop / conversions --------------------------------------------------------------------------------
def quantize_crop(lr_images, hr_images, hr_image_size, upscale_factor, hr_top, hr_left):
    """clamp(round(lr*255))/255 fused with the crop of both tensors (train_realesrnet.py:374-377).
    ALIASING: when the HR window is the whole tile (hr_top = hr_left = 0, hr_image_size = the tile's edge: every batch of the
    reference's configuration) the returned HR tensor IS `hr_images` (a float contiguous input is not copied either) -- the reference's
    random_crop always returns fresh tensors.  Callers that keep the pair beyond the caller's next write to that buffer must copy
    (degrade.Degrader / DegradationPrefetcher do)."""
    lr, hr = _img(lr_images, "quantize_crop"), _img(hr_images, "quantize_crop")
    b, c = lr.shape[:2]
    ls = hr_image_size // upscale_factor
    plr = torch.empty((b, c, ls, ls), dtype=torch.float32, device=lr.device)
    # an HR window that is the whole tile needs no copy: the target is only ever read (the reference's per-sample copy loop,
    # imgproc.py:1921-1932, exists to cut a window out)
    whole = hr_top == 0 and hr_left == 0 and hr_image_size == hr.shape[2] == hr.shape[3]
    phr = hr if whole else torch.empty((b, c, hr_image_size, hr_image_size), dtype=torch.float32, device=lr.device)
    _lib.check(_lib.lib().resr_quantize_crop(_lib.ptr(lr), _lib.ptr(hr), _lib.ptr(plr), None if whole else _lib.ptr(phr), b, c, lr.shape[2],
                                             lr.shape[3], hr.shape[2], hr.shape[3], hr_image_size, upscale_factor,
                                             hr_top, hr_left, _lib.stream_ptr(lr)), "resr_quantize_crop")
    return plr, phr


def random_crop(lr_images: torch.Tensor, hr_images: torch.Tensor, hr_image_size: int,
                upscale_factor: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference imgproc.py:1894-1934: ONE (top,left) for the whole batch from python `random`; the LR window
    starts at hr offset // upscale (up to upscale-1 HR pixels of misalignment, as in the reference).
    NOTE: the reference crops an already-quantised lr; this helper does not re-quantise."""
    hr_h, hr_w = hr_images.shape[2:]
    hr_top = random.randint(0, hr_h - hr_image_size)
    hr_left = random.randint(0, hr_w - hr_image_size)
    lt, ll, ls = hr_top // upscale_factor, hr_left // upscale_factor, hr_image_size // upscale_factor
    _lib.require_cuda(lr_images, "random_crop")
    return (lr_images[:, :, lt:lt + ls, ll:ll + ls].contiguous(),
            hr_images[:, :, hr_top:hr_top + hr_image_size, hr_left:hr_left + hr_image_size].contiguous())


def image_to_tensor(image: np.ndarray, range_norm: bool, half: bool) -> torch.Tensor:
    """HWC float ndarray -> CHW tensor (reference imgproc.py:1540-1567)."""
    t = torch.from_numpy(np.ascontiguousarray(image.transpose(2, 0, 1)))
    if range_norm:
        t = t.mul(2.0).sub(1.0)
    return t.half() if half else t


def tensor_to_image(tensor: torch.Tensor, range_norm: bool, half: bool) -> Any:
    """[1,C,H,W] in [0,1] -> HWC uint8, truncating like the reference (imgproc.py:1594)."""
    if range_norm:
        tensor = tensor.add(1.0).div(2.0)
    if half:
        tensor = tensor.half()
    return tensor.squeeze(0).permute(1, 2, 0).mul(255).clamp(0, 255).cpu().numpy().astype("uint8")


# ---- host kernel synthesis (float64 numpy) ---------------------------------------------------------------
def _quad_form(kernel_size: int, sigma_x: float, sigma_y: float, theta: float, isotropic: bool) -> np.ndarray:
    ax = np.arange(-kernel_size // 2 + 1.0, kernel_size // 2 + 1.0)
    xx, yy = np.meshgrid(ax, ax)
    grid = np.stack([xx, yy], axis=2)
    if isotropic:
        cov = np.array([[sigma_x ** 2, 0], [0, sigma_x ** 2]])
    else:
        rot = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
        cov = rot @ np.array([[sigma_x ** 2, 0], [0, sigma_y ** 2]]) @ rot.T
    return np.sum(np.dot(grid, np.linalg.inv(cov)) * grid, 2)


def _kernel_of(family: str, kernel_size, sigma_x, sigma_y, theta, beta, isotropic) -> np.ndarray:
    q = _quad_form(kernel_size, sigma_x, sigma_y, theta, isotropic)
    if family == "gaussian":
        k = np.exp(-0.5 * q)
    elif family == "generalized":
        k = np.exp(-0.5 * np.power(q, beta))
    else:
        k = np.reciprocal(np.power(q, beta) + 1)
    return k / np.sum(k)


def random_mixed_kernels(kernel_type: list, kernel_prob: Sequence[float], kernel_size: int, sigma_x_range: list,
                         sigma_y_range: list, rotation_range: list, generalized_kernel_beta_range: list,
                         plateau_kernel_beta_range: list, noise_range: tuple = None) -> np.ndarray:
    """Reference imgproc.py:492-573 (same draw order from `random` / `np.random`)."""
    kind = random.choices(kernel_type, kernel_prob)[0]
    family = "generalized" if kind.startswith("generalized") else ("plateau" if kind.startswith("plateau") else "gaussian")
    isotropic = not kind.endswith("anisotropic")
    sigma_x = np.random.uniform(sigma_x_range[0], sigma_x_range[1])
    if isotropic:
        sigma_y, theta = sigma_x, 0
    else:
        sigma_y = np.random.uniform(sigma_y_range[0], sigma_y_range[1])
        theta = np.random.uniform(rotation_range[0], rotation_range[1])
    beta = 1.0
    if family != "gaussian":
        rng = generalized_kernel_beta_range if family == "generalized" else plateau_kernel_beta_range
        beta = np.random.uniform(rng[0], 1) if np.random.uniform() < 0.5 else np.random.uniform(1, rng[1])
    k = _kernel_of(family, kernel_size, sigma_x, sigma_y, theta, beta, isotropic)
    if noise_range is not None and family != "plateau":
        k = k * np.random.uniform(noise_range[0], noise_range[1], size=k.shape)
    return k / np.sum(k)


def generate_sinc_kernel(cutoff: float, kernel_size: int, padding: int = 0) -> np.ndarray:
    """2-D sinc (jinc) low-pass kernel, reference imgproc.py:576-603."""
    from scipy import special
    assert kernel_size % 2 == 1, "Kernel size must be an odd number."
    c = (kernel_size - 1) / 2
    ii, jj = np.meshgrid(np.arange(kernel_size), np.arange(kernel_size), indexing="ij")
    r = np.sqrt((ii - c) ** 2 + (jj - c) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = cutoff * special.j1(cutoff * r) / (2 * np.pi * r)
    k[(kernel_size - 1) // 2, (kernel_size - 1) // 2] = cutoff ** 2 / (4 * np.pi)
    k = k / np.sum(k)
    if padding > kernel_size:
        p = (padding - kernel_size) // 2
        k = np.pad(k, ((p, p), (p, p)))
    return k


# ---------------------------------------------------------------------------------------------------------------
# Host side of the data path: file-side augmentation and the MATLAB-style resize used for the
# validation LR images.  CPU / numpy like the reference's DataLoader workers; no cv2.
# ---------------------------------------------------------------------------------------------------------------
def _cubic_kernel(x: np.ndarray) -> np.ndarray:
    """Keys cubic, a = -0.5 (reference imgproc.py:52-69), in the dtype of x."""
    ax = np.abs(x)
    ax2, ax3 = ax * ax, ax * ax * ax
    return ((1.5 * ax3 - 2.5 * ax2 + 1) * (ax <= 1) + (-0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2) * ((ax > 1) & (ax <= 2))).astype(x.dtype)


def _resize_matrix(in_length: int, out_length: int, scale: float, antialiasing: bool, dtype=np.float32) -> np.ndarray:
    """[out_length, in_length] matrix of the 1-D MATLAB `imresize` bicubic pass, symmetric edge replication folded in
    (reference imgproc.py:93-167 builds the same weights/indices in float32 and applies them row by row; NIQE's
    half-size pass, image_quality_assessment.py:520-591, is the same construction in float64)."""
    kernel_width = 4.0
    aa = scale < 1 and antialiasing
    if aa:
        kernel_width = kernel_width / scale
    f = dtype
    x = np.linspace(1, out_length, out_length, dtype=f)
    u = (x / f(scale) + f(0.5 * (1 - 1 / scale))).astype(f)
    left = np.floor(u - f(kernel_width / 2))
    p = math.ceil(kernel_width) + 2
    idx = left[:, None] + np.arange(p, dtype=f)[None, :]                     # 1-based input positions
    dist = (u[:, None] - idx).astype(f)
    w = f(scale) * _cubic_kernel(dist * f(scale)) if aa else _cubic_kernel(dist)
    w = (w / w.sum(1, keepdims=True)).astype(f)
    idx = idx.astype(np.int64)
    idx = np.where(idx < 1, 1 - idx, idx)                                     # symmetric: 0 -> 1, -1 -> 2, ...
    idx = np.where(idx > in_length, 2 * in_length + 1 - idx, idx)             # n+1 -> n, n+2 -> n-1, ...
    m = np.zeros((out_length, in_length), dtype=f)
    np.add.at(m, (np.repeat(np.arange(out_length), p), (idx - 1).reshape(-1)), w.reshape(-1))
    return m


def image_resize(image: Any, scale_factor: float, antialiasing: bool = True) -> Any:
    """MATLAB `imresize` (bicubic, optional antialiasing) -- reference imgproc.py:1599-1687.  numpy HWC / HW or
    torch CHW / HW in, same kind out, float32, not rounded, output size ceil(in * scale)."""
    is_np = isinstance(image, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(image)).float() if is_np else image.float()
    squeeze = t.ndim == 2
    if squeeze:
        t = t.unsqueeze(-1) if is_np else t.unsqueeze(0)
    if is_np:
        t = t.permute(2, 0, 1)
    _, in_h, in_w = t.shape
    out_h, out_w = math.ceil(in_h * scale_factor), math.ceil(in_w * scale_factor)
    mh = torch.from_numpy(_resize_matrix(in_h, out_h, scale_factor, antialiasing)).to(t.device)
    mw = torch.from_numpy(_resize_matrix(in_w, out_w, scale_factor, antialiasing)).to(t.device)
    out = torch.matmul(torch.matmul(mh, t), mw.t())                           # H pass, then W pass
    if is_np:
        out = out.permute(1, 2, 0)
        out = out.squeeze(-1) if squeeze else out
        return out.numpy()
    return out.squeeze(0) if squeeze else out


def center_crop(image: np.ndarray, image_size: int) -> np.ndarray:
    """Reference imgproc.py:1871-1891."""
    h, w = image.shape[:2]
    top, left = (h - image_size) // 2, (w - image_size) // 2
    return image[top:top + image_size, left:left + image_size, ...]


def _rotate_right_angle(image: np.ndarray, angle: int) -> np.ndarray:
    """What `cv2.warpAffine(image, cv2.getRotationMatrix2D((w//2, h//2), angle, 1), (w, h))` yields for the four
    right angles (reference imgproc.py:1937-1963): an exact pixel permutation about the *integer* centre
    (cx, cy) = (w//2, h//2), output size unchanged, pixels that fall outside are 0.  With an even side the centre is
    half a pixel off, so a 90/180/270 rotation loses one row/column and gains a black one.  (cv2 is absent from the
    image: pinned to the affine formula, unpinned vs OpenCV.)"""
    angle %= 360
    if angle == 0:
        return image
    h, w = image.shape[:2]
    cx, cy = w // 2, h // 2
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")           # destination coordinates
    if angle == 90:       # dst = [[0, 1, cx - cy], [-1, 0, cx + cy]] . src
        sx, sy = cx + cy - ys, xs - cx + cy
    elif angle == 180:
        sx, sy = 2 * cx - xs, 2 * cy - ys
    else:                 # 270
        sx, sy = ys - cy + cx, cx + cy - xs
    ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    out = np.zeros_like(image)
    out[ys[ok], xs[ok]] = image[sy[ok], sx[ok]]
    return out


def random_rotate(image: np.ndarray, angles: list, center: tuple = None, scale_factor: float = 1.0) -> np.ndarray:
    """Reference imgproc.py:1937-1963 for right angles about the default centre (the only use, dataset.py:70)."""
    angle = random.choice(angles)
    if center is not None or scale_factor != 1.0 or angle % 90:
        raise NotImplementedError("random_rotate: only right angles about the default centre (dataset.py:70)")
    return _rotate_right_angle(image, int(angle))


def random_horizontally_flip(image: np.ndarray, p: float) -> np.ndarray:
    """Reference imgproc.py:1966-1982 (`cv2.flip(image, 1)`)."""
    return image[:, ::-1].copy() if random.random() < p else image


def random_vertically_flip(image: np.ndarray, p: float) -> np.ndarray:
    """Reference imgproc.py:1985-2001 (`cv2.flip(image, 0)`)."""
    return image[::-1].copy() if random.random() < p else image


def rgb2ycbcr_torch(tensor: torch.Tensor, only_use_y_channel: bool) -> torch.Tensor:
    """ITU-R BT.601 as MATLAB `rgb2ycbcr`, [N,3,H,W] in [0,1] (reference imgproc.py:1815-1840)."""
    nhwc = tensor.permute(0, 2, 3, 1)
    if only_use_y_channel:
        w = torch.tensor([[65.481], [128.553], [24.966]], dtype=tensor.dtype, device=tensor.device)
        return (torch.matmul(nhwc, w).permute(0, 3, 1, 2) + 16.0) / 255.0
    m = torch.tensor([[65.481, -37.797, 112.0], [128.553, -74.203, -93.786], [24.966, 112.0, -18.214]],
                     dtype=tensor.dtype, device=tensor.device)
    b = torch.tensor([16.0, 128.0, 128.0], dtype=tensor.dtype, device=tensor.device).view(1, 3, 1, 1)
    return (torch.matmul(nhwc, m).permute(0, 3, 1, 2) + b) / 255.0


def read_image_rgb(path: str) -> np.ndarray:
    """File -> HWC float32 RGB in [0,1] (the reference reads BGR with cv2 and converts after the augmentation,
    dataset.py:66-75; the augmentations are channel-agnostic, so reading RGB directly is equivalent)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
