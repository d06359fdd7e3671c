"""Import the read-only reference (/root/reference) in a container without torchvision / cv2.

Only usable where /root/reference exists (the build container).  Never imported by the
`-m gpu` tests, smoke() or bench.py.  Three formulas are stand-ins for third-party
libraries the image lacks -- parity for them is "pinned to formula" (the file-side cv2 functions
of dataset.py -- imread, cvtColor, flip, getRotationMatrix2D, warpAffine -- are stand-ins too, textbook forms):
  * torchvision rgb_to_grayscale  -> 0.2989 R + 0.587 G + 0.114 B
  * cv2.getGaussianKernel(k, s<=0) -> sigma = 0.3*((k-1)*0.5-1)+0.8, exp(-x^2/2s^2) normalised
  * torchvision to_tensor          -> HWC float ndarray -> CHW tensor
Everything else executes the reference's own code on torch CPU.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("RESR_REFERENCE", "/root/reference")


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install_stubs():
    if "cv2" not in sys.modules:
        cv2 = _stub("cv2")

        def getGaussianKernel(ksize, sigma):
            if sigma <= 0:
                sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
            x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
            k = np.exp(-(x * x) / (2.0 * sigma * sigma))
            return (k / k.sum()).reshape(ksize, 1)

        cv2.getGaussianKernel = getGaussianKernel
        cv2.IMREAD_UNCHANGED = -1
        cv2.COLOR_BGR2RGB = 4
        cv2.COLOR_RGB2BGR = 4

        # file-side functions used by the reference's dataset.py (host data path goldens): generic textbook forms
        def imread(path, flags=-1):
            from PIL import Image
            with Image.open(path) as im:
                return np.asarray(im.convert("RGB"))[:, :, ::-1].copy()          # BGR uint8, like OpenCV

        def cvtColor(img, code):
            return np.ascontiguousarray(img[:, :, ::-1])

        def flip(img, code):
            return np.ascontiguousarray(img[:, ::-1] if code == 1 else img[::-1])

        def getRotationMatrix2D(center, angle, scale):
            a = np.deg2rad(angle)
            al, be = scale * np.cos(a), scale * np.sin(a)
            return np.array([[al, be, (1 - al) * center[0] - be * center[1]],
                             [-be, al, be * center[0] + (1 - al) * center[1]]], dtype=np.float64)

        def warpAffine(img, m, dsize):
            # dst(x, y) = src(M^-1 (x, y)), bilinear, constant 0 border (OpenCV defaults)
            w, h = dsize
            mi = np.linalg.inv(np.vstack([m, [0, 0, 1]]))
            ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
            sx = mi[0, 0] * xs + mi[0, 1] * ys + mi[0, 2]
            sy = mi[1, 0] * xs + mi[1, 1] * ys + mi[1, 2]
            sx, sy = np.round(sx * 1024) / 1024, np.round(sy * 1024) / 1024       # fixed-point coordinates
            x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
            fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
            src = img.astype(np.float64)
            if src.ndim == 2:
                src = src[..., None]

            def tap(yy, xx):
                ok = (xx >= 0) & (xx < src.shape[1]) & (yy >= 0) & (yy < src.shape[0])
                out = np.zeros((h, w, src.shape[2]))
                out[ok] = src[yy[ok], xx[ok]]
                return out

            out = (tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) +
                   tap(y0 + 1, x0) * (1 - fx) * fy + tap(y0 + 1, x0 + 1) * fx * fy)
            return out.astype(img.dtype).reshape(h, w, *img.shape[2:])

        cv2.imread, cv2.cvtColor, cv2.flip = imread, cvtColor, flip
        cv2.getRotationMatrix2D, cv2.warpAffine = getRotationMatrix2D, warpAffine
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tvm = _stub("torchvision.models")
        tvf = _stub("torchvision.models.feature_extraction")
        tvt = _stub("torchvision.transforms")
        tvtf = _stub("torchvision.transforms.functional")
        tvtt = _stub("torchvision.transforms.functional_tensor")
        tv.models, tv.transforms = tvm, tvt
        tvm.feature_extraction = tvf
        tvm.vgg19 = lambda *a, **k: None
        tvf.create_feature_extractor = lambda *a, **k: None
        tvt.Normalize = lambda *a, **k: None
        tvt.functional, tvt.functional_tensor = tvtf, tvtt

        def to_tensor(img):
            return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))

        def rgb_to_grayscale(img, num_output_channels=1):
            r, g, b = img.unbind(dim=-3)
            l = (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)
            if num_output_channels == 3:
                return l.expand(img.shape)
            return l

        tvtf.to_tensor = to_tensor
        tvtt.rgb_to_grayscale = rgb_to_grayscale


def load(name):
    """Import reference module `name` (model, imgproc, config, ...) by path."""
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    key = "resr_reference_" + name
    if key in sys.modules:
        return sys.modules[key]
    # config.py touches torch.backends.cudnn only; harmless on CPU
    mod = importlib.import_module(name)
    sys.modules[key] = mod
    return mod
