"""Oracle restatement of the INTEGER mode of blur / resize / JPEG (north_star: "blur/resize/JPEG bit-exact in integer mode";
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


# ---------------------------------------------------------------------------------------------------------------------
# JPEG round trip in integer mode: DiffJPEG(differentiable=False) (imgproc.py:1462-1494; colour :1195-1262, DCT :1296-1316,
# quantisation :1319-1353, inverse :1356-1459, tables :40-49, quality -> factor :1124-1141) on uint8 images with fixed-point
# constants and int64 accumulation.  Fixed-point formats (the HIP kernel csrc/degrade_int.hip:jpeg_u8_kernel must agree bit
# for bit; every ">>" is an arithmetic shift, i.e. floor; nothing exceeds 2^55):
#   colour matrices  rint(M * 2^20), samples level-shifted; luma x 4 and the 2x2 chroma SUM -> Q22
#   DCT matrix       C[u][x] = rint(0.5 * alpha(u) * cos((2x+1) u pi / 16) * 2^20); pass 1 descaled by 20 -> Q22, pass 2 -> Q42
#   quantiser step   S = max(1, rint(table * factor * 2^20)), factor in float64 from the float32 quality;
#                    q = round-half-even(F / (S * 2^22)), dequantised q * S (Q20)
#   inverse          the same matrix transposed, pass 1 descaled by 24 -> Q16, pass 2 by 20 -> Q16; colour back with
#                    rint(M^-1 * 2^20): out = clamp((v + 2^35) >> 36, 0, 255)
# ---------------------------------------------------------------------------------------------------------------------
_Y_STD = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51,
                   87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120,
                   101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float64).reshape(8, 8).T          # imgproc.py:40-44 (transposed)
_C_STD = np.full((8, 8), 99.0)
_C_STD[:4, :4] = np.array([17, 18, 24, 47, 18, 21, 26, 66, 24, 26, 56, 99, 47, 66, 99, 99], dtype=np.float64).reshape(4, 4).T  # :45-48
JQ = 20
_TO_YCC = np.rint(np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]]) * (1 << JQ)).astype(np.int64)  # :1217-1222
_R_CR, _G_CB, _G_CR, _B_CB = (int(np.rint(v * (1 << JQ))) for v in (1.402, -0.344136, -0.714136, 1.772))                         # :1435-1440


def jpeg_dct_matrix() -> np.ndarray:
    u = np.arange(8, dtype=np.float64)[:, None]
    x = np.arange(8, dtype=np.float64)[None, :]
    alpha = np.where(u == 0, 1.0 / np.sqrt(2.0), 1.0)
    return np.rint(0.5 * alpha * np.cos((2 * x + 1) * u * np.pi / 16) * (1 << JQ)).astype(np.int64)


def jpeg_steps(quality: np.ndarray) -> np.ndarray:
    """float32 quality [N] -> int64 [N,2,8,8] quantiser steps in Q20 (luma, chroma)."""
    q = np.asarray(quality, dtype=np.float32).astype(np.float64)
    factor = np.where(q < 50, 50.0 / q, 2.0 - q / 50.0)                       # imgproc.py:1134-1139: (5000/q)/100, (200-2q)/100
    tabs = np.stack([_Y_STD, _C_STD])[None] * factor[:, None, None, None]
    return np.maximum(np.rint(tabs * float(1 << JQ)), 1).astype(np.int64)


def _blocks(p: np.ndarray) -> np.ndarray:
    n, h, w = p.shape
    return p.reshape(n, h // 8, 8, w // 8, 8).transpose(0, 1, 3, 2, 4)           # [N, by, bx, 8, 8]


def _unblocks(b: np.ndarray) -> np.ndarray:
    n, by, bx = b.shape[:3]
    return b.transpose(0, 1, 3, 2, 4).reshape(n, by * 8, bx * 8)


def _div_half_even(a: np.ndarray, d: np.ndarray) -> np.ndarray:
    q, r = np.divmod(2 * a + d, 2 * d)                                            # floor((a + d/2) / d), exact
    return q - ((r == 0) & (q % 2 != 0))                                          # a tie went up: back to the even neighbour


def jpeg_u8(img: np.ndarray, quality: np.ndarray, return_coefficients: bool = False):
    """uint8 [N,3,H,W], float32 quality [N] -> uint8 [N,3,H,W] (and the quantised coefficients [N, by, bx, 8, 8] of Y, Cb, Cr)."""
    n, _, h, w = img.shape
    hp, wp = -h % 16, -w % 16
    x = np.pad(img.astype(np.int64), ((0, 0), (0, 0), (0, hp), (0, wp)))          # zero padding of the RGB image, imgproc.py:1486-1488
    ycc = np.einsum("kc,nchw->nkhw", _TO_YCC, x)                                  # Q20; Cb, Cr without their +128 (= level-shifted)
    ycc[:, 0] -= 128 << JQ
    planes = [ycc[:, 0] << 2]                                                     # Q22
    for k in (1, 2):                                                              # 2x2 chroma SUM = 4 x the average, :1240-1262
        c = ycc[:, k]
        planes.append(c[:, 0::2, 0::2] + c[:, 0::2, 1::2] + c[:, 1::2, 0::2] + c[:, 1::2, 1::2])
    C = jpeg_dct_matrix()
    steps = jpeg_steps(quality)
    out, coefs = [], []
    for k, p in enumerate(planes):
        b = _blocks(p)                                                            # [..., x, y]
        t = (np.einsum("ux,nabxy->nabuy", C, b) + (1 << 19)) >> 20                # Q22
        f = np.einsum("vy,nabuy->nabuv", C, t)                                    # Q42
        st = steps[:, min(k, 1)][:, None, None]
        q = _div_half_even(f, st << 22)
        coefs.append(q)
        s = (np.einsum("ux,nabuv->nabxv", C, q * st) + (1 << 23)) >> 24           # Q20 * Q20 -> Q16
        r = (np.einsum("vy,nabxv->nabxy", C, s) + (1 << 19)) >> 20                # Q16
        out.append(_unblocks(r))
    Y = (out[0] + (128 << 16)) << JQ                                              # Q36
    cb = np.repeat(np.repeat(out[1], 2, axis=1), 2, axis=2)                       # nearest x2, imgproc.py:1412-1432
    cr = np.repeat(np.repeat(out[2], 2, axis=1), 2, axis=2)
    rgb = np.stack([Y + _R_CR * cr, Y + _G_CB * cb + _G_CR * cr, Y + _B_CB * cb], axis=1)
    res = np.clip((rgb + (1 << 35)) >> 36, 0, 255).astype(np.uint8)[:, :, :h, :w]
    return (res, coefs) if return_coefficients else res
