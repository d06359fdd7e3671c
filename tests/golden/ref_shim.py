"""Import the read-only reference (/root/reference) in a container without torchvision / cv2.

Only usable where /root/reference exists (the build container).  Never imported by the
`-m gpu` tests, smoke() or bench.py.  Three formulas are stand-ins for third-party
libraries the image lacks -- parity for them is "pinned to formula":
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
