"""Oracle restatement of the reference's device-side degradation ops and host kernel synthesis
(reference: imgproc.py, dataset.py:82-143, train_realesrnet.py:262-377).  CPU fp32 torch / float64 numpy.

Randomness: the reference draws from three host RNGs and the torch device RNG.  Here every op takes
its random inputs explicitly (fields, per-sample scalars) *or* draws them from torch's global
generator in exactly the reference's order, so that a seeded reference call and a seeded oracle call
coincide (that is how the golden vectors pin these functions).
"""
from __future__ import annotations

import math
import random
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from scipy import special

# ---- JPEG tables: the reference stores the *transposed* standard tables (imgproc.py:40-49) -----
_Y_STD = np.array(
    [[16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55], [14, 13, 16, 24, 40, 57, 69, 56],
     [14, 17, 22, 29, 51, 87, 80, 62], [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
     [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]], dtype=np.float32)
Y_TABLE = torch.from_numpy(_Y_STD.T.copy())
_c = np.full((8, 8), 99, dtype=np.float32)
_c[:4, :4] = np.array([[17, 18, 24, 47], [18, 21, 26, 66], [24, 26, 56, 99], [47, 66, 99, 99]], dtype=np.float32).T
C_TABLE = torch.from_numpy(_c)


# ---- blur kernels ----------------------------------------------------------------------------------
def gaussian_kernel_1d(ksize: int, sigma: float = 0.0) -> np.ndarray:
    """cv2.getGaussianKernel (general branch): sigma <= 0 -> 0.3*((k-1)*0.5-1)+0.8; float64 [k]."""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def usm_kernel(radius: int = 50, sigma: float = 0.0) -> torch.Tensor:
    """USMSharp.__init__ buffer (imgproc.py:1516-1524): outer product in float64, stored fp32 [1,k,k]."""
    if radius % 2 == 0:
        radius += 1
    g = gaussian_kernel_1d(radius, sigma).reshape(-1, 1)
    return torch.from_numpy(np.dot(g, g.T).astype(np.float32)).unsqueeze(0)


def filter2d(img: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """filter2d_torch (imgproc.py:1089-1121): reflect pad k//2, cross-correlation; kernel [1,k,k] shared
    or [B,k,k] per sample."""
    k = kernel.shape[-1]
    if k % 2 != 1:
        raise ValueError("Wrong kernel size.")
    b, c, h, w = img.shape
    p = F.pad(img, (k // 2,) * 4, mode="reflect")
    if kernel.shape[0] == 1:
        return F.conv2d(p.reshape(b * c, 1, *p.shape[-2:]), kernel.reshape(1, 1, k, k)).reshape(b, c, h, w)
    wts = kernel.reshape(b, 1, k, k).repeat(1, c, 1, 1).reshape(b * c, 1, k, k)
    return F.conv2d(p.reshape(1, b * c, *p.shape[-2:]), wts, groups=b * c).reshape(b, c, h, w)


def usm_sharp(img: torch.Tensor, kernel: torch.Tensor, weight: float = 0.5, threshold: float = 10) -> torch.Tensor:
    """USMSharp.forward (imgproc.py:1526-1537)."""
    blur = filter2d(img, kernel)
    res = img - blur
    mask = (res.abs() * 255 > threshold).float()
    soft = filter2d(mask, kernel)
    sharp = (img + weight * res).clip(0, 1)
    return soft * sharp + (1 - soft) * img


# ---- noise (imgproc.py:829-1086) -------------------------------------------------------------------------
def _finish(out: torch.Tensor, clip: bool, rounds: bool) -> torch.Tensor:
    if clip and rounds:
        return torch.clamp((out * 255.0).round(), 0, 255) / 255.
    if clip:
        return torch.clamp(out, 0, 1)
    if rounds:
        return (out * 255.0).round() / 255.
    return out


def gaussian_noise_from_fields(img, sigma, gray, field_gray, field_color):
    """_generate_gaussian_noise_torch with the two randn draws injected: field_gray [h,w] (ONE field for
    the whole batch, imgproc.py:854-855), field_color [B,3,h,w]; sigma, gray: [B]."""
    b = img.shape[0]
    s = sigma.view(b, 1, 1, 1)
    gr = gray.view(b, 1, 1, 1)
    noise = field_color * s / 255.
    if gray.sum() > 0:
        ng = (field_gray * s / 255.).view(b, 1, *img.shape[2:])
        noise = noise * (1 - gr) + ng * gr
    return noise


def random_add_gaussian_noise(img, sigma_range, gray_prob, clip=True, rounds=False):
    """random_add_gaussian_noise_torch (imgproc.py:1029-1057), drawing from torch's global generator in
    the reference's order: sigma rand, gray rand, [gray randn], colour randn."""
    b = img.shape[0]
    sigma = torch.rand(b) * (sigma_range[1] - sigma_range[0]) + sigma_range[0]
    gray = (torch.rand(b) < gray_prob).float()
    fg = torch.randn(*img.shape[2:]) if gray.sum() > 0 else torch.zeros(img.shape[2:])
    fc = torch.randn(*img.shape)
    return _finish(img + gaussian_noise_from_fields(img, sigma, gray, fg, fc), clip, rounds)


def rgb_to_gray(img: torch.Tensor) -> torch.Tensor:
    """torchvision rgb_to_grayscale formula (stand-in pinned to formula)."""
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def poisson_vals(img_q: torch.Tensor) -> torch.Tensor:
    """2^ceil(log2(#unique values)) per sample (imgproc.py:892-894, 903-905); img_q already on the k/255 grid."""
    counts = [len(torch.unique(img_q[i])) for i in range(img_q.shape[0])]
    return img_q.new_tensor([2 ** np.ceil(np.log2(c)) for c in counts]).view(-1, 1, 1, 1)


def poisson_noise(img, scale, gray, sampler=torch.poisson):
    """_generate_poisson_noise_torch (imgproc.py:866-916) with the sampler injectable."""
    b, _, h, w = img.shape
    gr = gray.view(b, 1, 1, 1)
    use_gray = gray.sum() > 0
    if use_gray:
        g = torch.clamp((rgb_to_gray(img) * 255.0).round(), 0, 255) / 255.
        vg = poisson_vals(g)
        ng = (sampler(g * vg) / vg - g).expand(b, 3, h, w)
    q = torch.clamp((img * 255.0).round(), 0, 255) / 255.
    v = poisson_vals(q)
    noise = sampler(q * v) / v - q
    if use_gray:
        noise = noise * (1 - gr) + ng * gr
    return noise * scale.view(b, 1, 1, 1)


def random_add_poisson_noise(img, scale_range, gray_prob, clip=True, rounds=False):
    """random_add_poisson_noise_torch (imgproc.py:1060-1086), global-generator draw order as the reference."""
    b = img.shape[0]
    scale = torch.rand(b) * (scale_range[1] - scale_range[0]) + scale_range[0]
    gray = (torch.rand(b) < gray_prob).float()
    return _finish(img + poisson_noise(img, scale, gray), clip, rounds)


# ---- DiffJPEG(differentiable=False) (imgproc.py:1124-1141, 1195-1494) -----------------------------------
def quality_to_factor(q: torch.Tensor) -> torch.Tensor:
    """_calculate_quality_factor, vectorised: q<50 -> 50/q, else 2 - q/50 (as (200-2q)/100)."""
    q = q.float()
    return torch.where(q < 50, (5000.0 / q) / 100.0, (200.0 - q * 2) / 100.0)


def _dct_basis():
    t = np.zeros((8, 8, 8, 8), dtype=np.float32)
    for x in range(8):
        for y in range(8):
            for u in range(8):
                for v in range(8):
                    t[x, y, u, v] = np.cos((2 * x + 1) * u * np.pi / 16) * np.cos((2 * y + 1) * v * np.pi / 16)
    a = np.array([1.0 / np.sqrt(2)] + [1] * 7)
    return torch.from_numpy(t), torch.from_numpy((np.outer(a, a) * 0.25).astype(np.float32)), \
        torch.from_numpy(np.outer(a, a).astype(np.float32))


_DCT_T, _DCT_SCALE, _IDCT_ALPHA = _dct_basis()
_IDCT_T = _DCT_T.permute(2, 3, 0, 1).contiguous()   # imgproc.py:1361-1364: tensor[x,y,u,v] = cos((2u+1)x..)cos((2v+1)y..)
_RGB2YCC = torch.tensor([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]]).t()
_YCC2RGB = torch.tensor([[1.0, 0.0, 1.402], [1, -0.344136, -0.714136], [1, 1.772, 0]]).t()


def _blocks(plane: torch.Tensor) -> torch.Tensor:           # [B,H,W] -> [B, H/8*W/8, 8, 8]
    b, h, w = plane.shape
    return plane.view(b, h // 8, 8, w // 8, 8).permute(0, 1, 3, 2, 4).reshape(b, -1, 8, 8)


def _unblocks(blk: torch.Tensor, h: int, w: int) -> torch.Tensor:
    b = blk.shape[0]
    return blk.view(b, h // 8, w // 8, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h, w)


def diff_jpeg(img: torch.Tensor, quality: torch.Tensor, return_coeffs: bool = False):
    """DiffJPEG(False).forward for a [B] quality tensor (not mutated here)."""
    b, _, h, w = img.shape
    factor = quality_to_factor(quality).view(b, 1, 1, 1)
    hp, wp = (16 - h % 16) % 16, (16 - w % 16) % 16
    x = F.pad(img, (0, wp, 0, hp)) * 255
    H, W = h + hp, w + wp
    ycc = torch.tensordot(x.permute(0, 2, 3, 1), _RGB2YCC, dims=1) + torch.tensor([0.0, 128.0, 128.0])
    y = ycc[..., 0]
    cb = F.avg_pool2d(ycc[..., 1].unsqueeze(1), 2).squeeze(1)
    cr = F.avg_pool2d(ycc[..., 2].unsqueeze(1), 2).squeeze(1)
    coeffs, planes = {}, {}
    for name, plane, table in (("y", y, Y_TABLE), ("cb", cb, C_TABLE), ("cr", cr, C_TABLE)):
        blk = _blocks(plane) - 128
        dct = _DCT_SCALE * torch.tensordot(blk, _DCT_T, dims=2)
        q = torch.round(dct / (table.expand(b, 1, 8, 8) * factor))
        coeffs[name] = q
        deq = q * (table.expand(b, 1, 8, 8) * factor)
        rec = 0.25 * torch.tensordot(deq * _IDCT_ALPHA, _IDCT_T, dims=2) + 128
        ph, pw = (H, W) if name == "y" else (H // 2, W // 2)
        planes[name] = _unblocks(rec, ph, pw)
    up = lambda t: t.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    ycc2 = torch.stack([planes["y"], up(planes["cb"]), up(planes["cr"])], dim=3)
    rgb = torch.tensordot(ycc2 + torch.tensor([0.0, -128.0, -128.0]), _YCC2RGB, dims=1).permute(0, 3, 1, 2)
    out = (rgb.clamp(0, 255) / 255)[:, :, :h, :w]
    return (out, coeffs) if return_coeffs else out


# ---- final quantise + crop (train_realesrnet.py:374-377, imgproc.py:1894-1934) -----------------------------
def quantize(img: torch.Tensor) -> torch.Tensor:
    return torch.clamp((img * 255.0).round(), 0, 255) / 255.


def crop_pair(lr, hr, hr_size, upscale, hr_top, hr_left):
    """random_crop with the two random.randint draws injected; lr offset = hr offset // upscale."""
    lt, ll, ls = hr_top // upscale, hr_left // upscale, hr_size // upscale
    return lr[:, :, lt:lt + ls, ll:ll + ls].clone(), hr[:, :, hr_top:hr_top + hr_size, hr_left:hr_left + hr_size].clone()


# ---- host kernel synthesis (imgproc.py:72-90, 170-603; dataset.py:82-143), float64 numpy ---------------------
def _grid(k: int):
    ax = np.arange(-k // 2 + 1.0, k // 2 + 1.0)
    xx, yy = np.meshgrid(ax, ax)
    return np.stack([xx, yy], axis=2)


def _sigma_matrix(sx, sy, theta, isotropic):
    if isotropic:
        return np.array([[sx ** 2, 0], [0, sx ** 2]])
    d = np.array([[sx ** 2, 0], [0, sy ** 2]])
    u = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
    return u @ d @ u.T


def bivariate_kernel(kind: str, k: int, sx: float, sy: float, theta: float, beta: float = 1.0,
                     isotropic: bool = True) -> np.ndarray:
    """kind in {"gaussian", "generalized", "plateau"} (imgproc.py:225-327)."""
    g = _grid(k)
    inv = np.linalg.inv(_sigma_matrix(sx, sy, theta, isotropic))
    quad = np.sum(np.dot(g, inv) * g, 2)
    if kind == "gaussian":
        ker = np.exp(-0.5 * quad)
    elif kind == "generalized":
        ker = np.exp(-0.5 * np.power(quad, beta))
    elif kind == "plateau":
        ker = np.reciprocal(np.power(quad, beta) + 1)
    else:
        raise ValueError(kind)
    return ker / np.sum(ker)


def sinc_kernel(cutoff: float, k: int, pad_to: int = 0) -> np.ndarray:
    """generate_sinc_kernel (imgproc.py:576-603)."""
    c = (k - 1) / 2
    yy, xx = np.meshgrid(np.arange(k), np.arange(k), indexing="ij")
    r = np.sqrt((yy - c) ** 2 + (xx - c) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        ker = cutoff * special.j1(cutoff * r) / (2 * np.pi * r)
    ker[(k - 1) // 2, (k - 1) // 2] = cutoff ** 2 / (4 * np.pi)
    ker = ker / np.sum(ker)
    if pad_to > k:
        p = (pad_to - k) // 2
        ker = np.pad(ker, ((p, p), (p, p)))
    return ker


KERNEL_TYPES = ["isotropic", "anisotropic", "generalized_isotropic", "generalized_anisotropic",
                "plateau_isotropic", "plateau_anisotropic"]


def random_mixed_kernel(probs: Sequence[float], k: int, sigma_range, rot_range, gen_beta_range, plat_beta_range) -> np.ndarray:
    """random_mixed_kernels with noise_range=None (imgproc.py:492-573), same host-RNG draw order:
    random.choices (type), np.random.uniform (sigma_x[, sigma_y, rotation][, coin, beta])."""
    kind = random.choices(KERNEL_TYPES, probs)[0]
    iso = kind.endswith("isotropic") and not kind.endswith("anisotropic")
    sx = np.random.uniform(sigma_range[0], sigma_range[1])
    if iso:
        sy, th = sx, 0
    else:
        sy = np.random.uniform(sigma_range[0], sigma_range[1])
        th = np.random.uniform(rot_range[0], rot_range[1])
    if kind.startswith("generalized") or kind.startswith("plateau"):
        br = gen_beta_range if kind.startswith("generalized") else plat_beta_range
        beta = np.random.uniform(br[0], 1) if np.random.uniform() < 0.5 else np.random.uniform(1, br[1])
        return bivariate_kernel("generalized" if kind.startswith("generalized") else "plateau", k, sx, sy, th, beta, iso)
    return bivariate_kernel("gaussian", k, sx, sy, th, 1.0, iso)


def sample_sample_kernels(P: dict) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The three 21x21 kernels of one training sample (dataset.py:82-143), P = degradation_model_parameters_dict."""
    out = []
    for idx in ("1", "2"):
        ks = random.choice(P["gaussian_kernel_range"])
        if np.random.uniform() < P["sinc_kernel_probability" + idx]:
            lo = np.pi / 3 if ks < int(np.median(P["gaussian_kernel_range"])) else np.pi / 5
            ker = sinc_kernel(np.random.uniform(lo, np.pi), ks, 0)
        else:
            ker = random_mixed_kernel(P["gaussian_kernel_probability" + idx], ks, P["gaussian_sigma_range" + idx],
                                      [-math.pi, math.pi], P["generalized_kernel_beta_range" + idx],
                                      P["plateau_kernel_beta_range" + idx])
        p = (P["gaussian_kernel_range"][-1] - ks) // 2
        out.append(np.pad(ker, ((p, p), (p, p))))
    if np.random.uniform() < P["sinc_kernel_probability3"]:
        ks = random.choice(P["gaussian_kernel_range"])
        final = sinc_kernel(np.random.uniform(np.pi / 3, np.pi), ks, P["sinc_kernel_size"])
    else:
        final = np.zeros((P["sinc_kernel_size"],) * 2)
        final[P["sinc_kernel_size"] // 2, P["sinc_kernel_size"] // 2] = 1
    return out[0].astype(np.float32), out[1].astype(np.float32), final.astype(np.float32)
