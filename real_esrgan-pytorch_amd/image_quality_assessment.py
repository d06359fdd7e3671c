"""No-reference quality metric of the reference's validation loop: `NIQE`, the module
`train_realesrnet.py:102,430-466` / `train_realesrgan.py` evaluate on every validation batch
(reference image_quality_assessment.py:803-1032, MATLAB NIQE semantics).

Plain torch in float64 on whatever device the SR batch lives on (a handful of 7x7 filters and per-block moment
statistics: control-plane work, not a kernel target).  The prior (mean / covariance of the 36 natural-scene features)
is the reference's `results/pretrained_models/niqe_model.mat`; a copy of that data file is kept as a test fixture
(`tests/golden/niqe_model.mat`).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import imgproc

__all__ = ["NIQE", "niqe"]


def _gaussian_window(size: int = 7, sigma: float = 7.0 / 6.0) -> torch.Tensor:
    """MATLAB fspecial('gaussian'); the reference stores it as float32 before use (iqa.py:215-242)."""
    m = (size - 1) / 2.0
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    h = np.exp(-(x * x + y * y) / (2.0 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    h /= h.sum()
    return torch.from_numpy(h.astype(np.float32))


def _aggd(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Asymmetric generalised Gaussian fit of every block x[n,1,h,w] by moment matching on the grid 0.2:0.001:10
    (iqa.py:803-851, the `get_sigma=True` form): shape alpha, left / right scale."""
    grid = torch.arange(0.2, 10 + 0.001, 0.001).to(x)                       # float32 grid values, as in the reference
    r_gam = (2 * torch.lgamma(2.0 / grid) - (torch.lgamma(1.0 / grid) + torch.lgamma(3.0 / grid))).exp()
    neg, pos = x < 0, x > 0
    n_neg = neg.sum(dim=(-1, -2), dtype=torch.float32)
    n_pos = pos.sum(dim=(-1, -2), dtype=torch.float32)
    left = torch.sqrt((x * neg).pow(2).sum(dim=(-1, -2)) / (n_neg + 1e-8))
    right = torch.sqrt((x * pos).pow(2).sum(dim=(-1, -2)) / (n_pos + 1e-8))
    g = left / right
    rhat = x.abs().mean(dim=(-1, -2)).pow(2) / x.pow(2).mean(dim=(-1, -2))
    rhat_norm = rhat * (g.pow(3) + 1) * (g + 1) / (g.pow(2) + 1).pow(2)
    alpha = grid[(r_gam.unsqueeze(0) - rhat_norm).abs().argmin(dim=-1)]     # [n]
    scale = (torch.lgamma(1 / alpha) - torch.lgamma(3 / alpha)).exp().sqrt()
    return alpha, left.squeeze(-1) * scale, right.squeeze(-1) * scale


def _block_features(blocks: torch.Tensor) -> torch.Tensor:
    """18 features of every MSCN block [n,1,h,w] (iqa.py:854-883): AGGD of the block and of its products with the four
    circularly shifted copies (right, down, both diagonals)."""
    alpha, bl, br = _aggd(blocks)
    feats: List[torch.Tensor] = [alpha, (bl + br) / 2]
    for shift in ((0, 1), (1, 0), (1, 1), (1, -1)):
        alpha, bl, br = _aggd(blocks * torch.roll(blocks, shift, dims=(2, 3)))
        mean = (br - bl) * (torch.lgamma(2 / alpha) - torch.lgamma(1 / alpha)).exp()
        feats += [alpha, mean, bl, br]
    return torch.stack(feats, dim=-1)                                       # [n, 18]


def _resize_half(y: torch.Tensor) -> torch.Tensor:
    """MATLAB imresize(., 0.5) with antialiasing, in the tensor's own precision (iqa.py:726-800 for scale 0.5)."""
    _, _, h, w = y.shape
    mh = torch.from_numpy(imgproc._resize_matrix(h, math.ceil(h * 0.5), 0.5, True, np.float64)).to(y)
    mw = torch.from_numpy(imgproc._resize_matrix(w, math.ceil(w * 0.5), 0.5, True, np.float64)).to(y)
    return torch.matmul(torch.matmul(mh, y), mw.t())


def niqe(tensor: torch.Tensor, crop_border: int, mu_prior: torch.Tensor, cov_prior: torch.Tensor,
         block_size_height: int = 96, block_size_width: int = 96) -> torch.Tensor:
    """NIQE score of every image of an RGB batch [B,3,H,W] in [0,1] (iqa.py:886-998)."""
    if crop_border > 0:
        tensor = tensor[:, :, crop_border:-crop_border, crop_border:-crop_border]
    y = (imgproc.rgb2ycbcr_torch(tensor.float(), only_use_y_channel=True) * 255.0).round().to(torch.float64)
    b, _, h, w = y.shape
    nbh, nbw = h // block_size_height, w // block_size_width
    if nbh == 0 or nbw == 0:
        raise ValueError(f"NIQE needs at least one {block_size_height}x{block_size_width} block, got {h}x{w}")
    y = y[..., :nbh * block_size_height, :nbw * block_size_width]
    win = _gaussian_window().to(y).view(1, 1, 7, 7)
    feats = []
    for scale in (1, 2):
        mu = F.conv2d(F.pad(y, (3, 3, 3, 3), mode="replicate"), win)
        sq = F.conv2d(F.pad(y * y, (3, 3, 3, 3), mode="replicate"), win)
        sigma = torch.sqrt((sq - mu * mu).abs() + 1e-8)
        mscn = (y - mu) / (sigma + 1)
        bh, bw = block_size_height // scale, block_size_width // scale
        blocks = F.unfold(mscn, (bh, bw), stride=(bh, bw))                  # [b, bh*bw, nblocks]
        nblk = blocks.shape[-1]
        blocks = blocks.transpose(1, 2).reshape(b * nblk, 1, bh, bw)
        feats.append(_block_features(blocks).reshape(b, nblk, 18))
        if scale == 1:
            y = _resize_half(y / 255.0) * 255.0
    dist = torch.cat(feats, dim=-1)                                         # [b, nblocks, 36]
    scores = []
    for i in range(b):                                                      # blocks with a NaN feature are dropped
        rows = dist[i]
        nan = torch.isnan(rows)
        mu_d = torch.where(nan, torch.zeros_like(rows), rows).sum(0) / (~nan).to(rows.dtype).sum(0)
        ok = rows[~nan.any(dim=1)]
        c = ok - ok.mean(dim=0, keepdim=True)
        cov_d = c.t() @ c / (ok.shape[0] - 1)
        diff = (mu_prior.to(rows) - mu_d).unsqueeze(0)
        inv = torch.linalg.pinv((cov_prior.to(rows) + cov_d) / 2)
        scores.append(torch.sqrt((diff @ inv @ diff.t()).squeeze()))
    return torch.stack(scores).squeeze()


class NIQE(nn.Module):
    """Same constructor and call as the reference's `NIQE` (iqa.py:1001-1032): `NIQE(crop_border, niqe_model_path)`,
    `forward(sr) -> score` (scalar for a batch of one, else one score per image)."""

    def __init__(self, crop_border: int, niqe_model_path: str, block_size_height: int = 96,
                 block_size_width: int = 96) -> None:
        super().__init__()
        self.crop_border = crop_border
        self.niqe_model_path = niqe_model_path
        self.block_size_height = block_size_height
        self.block_size_width = block_size_width
        import scipy.io
        model = scipy.io.loadmat(niqe_model_path)
        self.register_buffer("mu_prisparam", torch.from_numpy(np.ravel(model["mu_prisparam"]).astype(np.float64)), persistent=False)
        self.register_buffer("cov_prisparam", torch.from_numpy(np.asarray(model["cov_prisparam"], dtype=np.float64)), persistent=False)

    def forward(self, raw_tensor: torch.Tensor) -> torch.Tensor:
        # the reference moves the priors to the image's dtype first (iqa.py:980-984)
        mu = self.mu_prisparam.to(raw_tensor.dtype).to(torch.float64)
        cov = self.cov_prisparam.to(raw_tensor.dtype).to(torch.float64)
        return niqe(raw_tensor, self.crop_border, mu, cov, self.block_size_height, self.block_size_width)
