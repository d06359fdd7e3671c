"""Oracle restatement of the INTEGER mode of blur / resize (north_star: "blur/resize/JPEG bit-exact in integer mode";
SURVEY.md §7 defines the mode: uint8-quantised inputs, fixed-point taps, integer accumulation).

TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).  Pure numpy integer arithmetic, written independently of
the product's host code: the HIP kernels (csrc/degrade_int.hip) must reproduce these results BIT FOR BIT, and both stay
within 1 LSB of the reference's float ops on the same uint8 inputs (imgproc.py:1089-1121 filter2d_torch; F.interpolate
at train_realesrnet.py:288,326-329,349-351,366-368), which tests/test_gpu_int_mode.py measures against the goldens.
Parity status: the integer mode is net-new (the reference has no integer path), so it is pinned to the reference only
through that <= 1 LSB distance to the reference's own float outputs (tests/golden/imgproc_filter.npz, imgproc_resize.npz).
"""
from __future__ import annotations

import numpy as np

QK, QR = 14, 11      # fractional bits of blur taps / resize weights


def _fix_sum(q: np.ndarray, want: np.ndarray) -> np.ndarray:
    """Put the rounding residue of each row on its largest-magnitude entry (first one on ties)."""
    q = q.copy()
    peak = np.abs(q).argmax(axis=1)
    q[np.arange(q.shape[0]), peak] += want - q.sum(axis=1)
    return q


def quantize_kernel(k: np.ndarray) -> np.ndarray:
    """float [..., kh, kw] -> int32 Q14 taps, each kernel summing to rint(sum * 2^14)."""
    k = np.asarray(k, dtype=np.float64)
    rows = k.reshape(-1, k.shape[-2] * k.shape[-1])
    q = _fix_sum(np.rint(rows * (1 << QK)).astype(np.int64), np.rint(rows.sum(axis=1) * (1 << QK)).astype(np.int64))
    return q.reshape(k.shape).astype(np.int32)


def filter2d_u8(img: np.ndarray, taps: np.ndarray) -> np.ndarray:
    """uint8 [N,C,H,W], int32 taps [1|N,kh,kw]: reflect pad (numpy 'reflect' == torch 'reflect'), correlation,
    (acc + 2^13) >> 14 with floor semantics, clamp to [0,255]."""
    n, c, h, w = img.shape
    kh, kw = taps.shape[-2:]
    ry, rx = kh // 2, kw // 2
    pad = np.pad(img.astype(np.int64), ((0, 0), (0, 0), (ry, ry), (rx, rx)), mode="reflect")
    acc = np.zeros((n, c, h, w), dtype=np.int64)
    for dy in range(kh):
        for dx in range(kw):
            t = taps[:, dy, dx].astype(np.int64).reshape(-1, 1, 1, 1)        # [1] or [N]: broadcasts over the batch
            acc += t * pad[:, :, dy:dy + h, dx:dx + w]
    return np.clip((acc + (1 << (QK - 1))) >> QK, 0, 255).astype(np.uint8)


def _cubic(t: np.ndarray):
    A = -0.75
    c1 = lambda x: ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0            # noqa: E731  |x| <= 1
    c2 = lambda x: ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A     # noqa: E731  1 < |x| < 2
    return np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], axis=1)


def axis_tables(n_in: int, n_out: int, scale, mode: str):
    """ATen's align_corners=False coordinate map in float64 -> (clamped indices [out,taps], Q11 weights [out,taps])."""
    s = (1.0 / scale) if scale else n_in / n_out
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * s - 0.5
    if mode == "bilinear":
        src = np.maximum(src, 0.0)
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        lam = src - i0
        idx = np.stack([i0, np.minimum(i0 + 1, n_in - 1)], axis=1)
        w = np.stack([1.0 - lam, lam], axis=1)
    else:
        i0 = np.floor(src).astype(np.int64)
        w = _cubic(src - i0)
        idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)
    q = _fix_sum(np.rint(w * (1 << QR)).astype(np.int64), np.full(n_out, 1 << QR, dtype=np.int64))
    return idx.astype(np.int64), q


def resize_u8(img: np.ndarray, out_hw, scale, mode: str) -> np.ndarray:
    """uint8 [N,C,H,W] -> uint8 [N,C,oh,ow]; `scale` = the scale_factor the caller gave, or None for size= semantics."""
    n, c, h, w = img.shape
    oh, ow = out_hw
    x = img.astype(np.int64)
    if mode == "area":      # adaptive_avg_pool2d windows, mean rounded half up
        out = np.empty((n, c, oh, ow), dtype=np.uint8)
        for oy in range(oh):
            y0, y1 = (oy * h) // oh, -((-(oy + 1) * h) // oh)
            for ox in range(ow):
                x0, x1 = (ox * w) // ow, -((-(ox + 1) * w) // ow)
                s = x[:, :, y0:y1, x0:x1].sum(axis=(2, 3))
                cnt = (y1 - y0) * (x1 - x0)
                out[:, :, oy, ox] = ((2 * s + cnt) // (2 * cnt)).astype(np.uint8)
        return out
    iy, wy = axis_tables(h, oh, scale, mode)
    ix, wx = axis_tables(w, ow, scale, mode)
    rows = (x[:, :, :, ix] * wx[None, None, None, :, :]).sum(axis=4)               # [N,C,H,ow]   horizontal pass
    acc = (rows[:, :, iy, :] * wy[None, None, :, :, None]).sum(axis=3)             # [N,C,oh,ow]  vertical pass
    return np.clip((acc + (1 << (2 * QR - 1))) >> (2 * QR), 0, 255).astype(np.uint8)
