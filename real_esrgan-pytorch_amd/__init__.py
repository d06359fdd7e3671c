"""real_esrgan-pytorch_amd -- the MI355X (gfx950) native Real-ESRGAN hot path.

Import as `real_esrgan_pytorch_amd` (the sibling shim package maps the importable name onto this
directory, whose name is fixed by the project layout and is not a valid Python identifier).

Contents: `model` (Generator / EMA behind the reference's nn.Module surface), `_lib` (ctypes
binding of csrc/libresr_hip.so, C-ABI in include/resr.h), `csrc/` (HIP kernels + the C-ABI).
"""
from . import _lib  # noqa: F401
from .model import EMA, Generator, ResidualDenseBlock, ResidualResidualDenseBlock  # noqa: F401
from .discriminator import Discriminator  # noqa: F401
from .content_loss import ContentLoss  # noqa: F401

__all__ = ["EMA", "Generator", "Discriminator", "ContentLoss", "ResidualDenseBlock", "ResidualResidualDenseBlock"]
