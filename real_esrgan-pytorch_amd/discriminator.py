"""U-Net discriminator with spectral norm (reference model.py:135-203) on the gfx950 kernels.

Same constructor, forward semantics ([N,3,H,W] -> [N,1,H,W] logits, H,W divisible by 8) and state_dict
keys as the reference (`conv1.*`, `{down_block1..3,up_block1..3,conv2,conv3}.0.{weight_orig,weight_u,
weight_v}`, `conv4.*`): the parameter containers are literally `spectral_norm(nn.Conv2d(...))` objects
built in the reference's order, so the init RNG stream is identical; their Python forward is never used.

forward / backward are ONE C-ABI call each (`resr_discriminator_forward` / `_backward`, include/resr.h;
plan and launch order in csrc/disc_native.hip): spectral norm (one power iteration per training-mode
forward, u / v updated in place like torch's hook, 1/sigma folded into the packed weights on the device),
MFMA implicit-GEMM convolutions with all 64-channel output groups of a layer in one launch, the three
4x4 / stride-2 convolutions as sparse-tap 3x3 convolutions over the space-to-depth image, bilinear x2,
skip adds and LeakyReLU masks; backward mirrors it and maps the weight gradients back to `weight_orig`.
Parameters are views into one flat fp32 arena, the spectral-norm vectors into another; every call works
inside one pooled workspace (activations, this call's u / v / sigma and packed weights, scratch).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch
from torch import nn
from torch.nn.utils import spectral_norm

from . import _lib
from .model import _precision_to_dtype


class _DWorkspace:
    """One workspace + the pack table that points into it; `busy` while a graph that saved into it is alive."""

    def __init__(self, nbytes: int, device, desc) -> None:
        L = _lib.lib()
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.buf[:256].zero_()     # 1 / sigma + the gradient pre-scale slot with its sticky back-off words: zero once (include/resr.h)
        n = int(L.resr_discriminator_pack_table(C.byref(desc), _lib.ptr(self.buf), None, 0))
        host = (_lib.PackChunk * n)()
        got = int(L.resr_discriminator_pack_table(C.byref(desc), _lib.ptr(self.buf), C.cast(host, C.c_void_p), n))
        if got != n:
            _lib.check(got if got < 0 else -1, "resr_discriminator_pack_table")
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(device)
        self.n_chunks = n
        self.busy = False
        self.owner = 0

    def acquire(self) -> int:
        self.owner += 1
        self.busy = True
        return self.owner

    def release(self, owner: int) -> None:
        if owner == self.owner:
            self.busy = False


class _DToken:
    def __init__(self, ws: _DWorkspace, owner: int) -> None:
        self.ws, self.owner = ws, owner

    def __del__(self) -> None:
        self.ws.release(self.owner)


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "Discriminator", training: bool, x: torch.Tensor, *params: torch.Tensor):
        y, desc, ws = module._run_forward(x, training)
        ctx.module, ctx.desc, ctx.ws = module, desc, ws
        ctx.x_needs_grad = x.requires_grad
        ctx.param_needs = [p.requires_grad for p in params]
        ctx.owner = 0
        if training:
            ctx.owner = ws.acquire()
            ctx._token = _DToken(ws, ctx.owner)
        return y

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        grads, gx = ctx.module._run_backward(ctx.desc, ctx.ws, gy.contiguous().float(), ctx.x_needs_grad, any(ctx.param_needs))
        ctx.ws.release(ctx.owner)
        return (None, None, gx) + tuple(g if need else None for g, need in zip(grads, ctx.param_needs))


class Discriminator(nn.Module):
    """Reference `Discriminator()` (model.py:135-203).  Extra keyword `precision`, as for `Generator`: "fast" (f16 MFMA, fp32
    accumulate), "exact16" (split-operand f16 MFMA on hi/lo pairs: fp32-class results, the mode that meets the 1e-3 parity
    tolerance) or "strict" (f32 MFMA); anything else raises."""

    def __init__(self, precision: Optional[str] = None) -> None:
        super().__init__()
        self.precision = precision or os.environ.get("RESR_PRECISION", "fast")
        self._dtype = _precision_to_dtype(self.precision)          # ValueError for an unknown precision: never a silent downgrade
        self.conv1 = nn.Conv2d(3, 64, (3, 3), (1, 1), (1, 1))
        self.down_block1 = nn.Sequential(spectral_norm(nn.Conv2d(64, 128, (4, 4), (2, 2), (1, 1), bias=False)), nn.LeakyReLU(0.2, True))
        self.down_block2 = nn.Sequential(spectral_norm(nn.Conv2d(128, 256, (4, 4), (2, 2), (1, 1), bias=False)), nn.LeakyReLU(0.2, True))
        self.down_block3 = nn.Sequential(spectral_norm(nn.Conv2d(256, 512, (4, 4), (2, 2), (1, 1), bias=False)), nn.LeakyReLU(0.2, True))
        self.up_block1 = nn.Sequential(spectral_norm(nn.Conv2d(512, 256, (3, 3), (1, 1), (1, 1), bias=False)), nn.LeakyReLU(0.2, True))
        self.up_block2 = nn.Sequential(spectral_norm(nn.Conv2d(256, 128, (3, 3), (1, 1), (1, 1), bias=False)), nn.LeakyReLU(0.2, True))
        self.up_block3 = nn.Sequential(spectral_norm(nn.Conv2d(128, 64, (3, 3), (1, 1), (1, 1), bias=False)), nn.LeakyReLU(0.2, True))
        self.conv2 = nn.Sequential(spectral_norm(nn.Conv2d(64, 64, (3, 3), (1, 1), (1, 1), bias=False)), nn.LeakyReLU(0.2, True))
        self.conv3 = nn.Sequential(spectral_norm(nn.Conv2d(64, 64, (3, 3), (1, 1), (1, 1), bias=False)), nn.LeakyReLU(0.2, True))
        self.conv4 = nn.Conv2d(64, 1, (3, 3), (1, 1), (1, 1))
        self._flat: Optional[torch.Tensor] = None       # fp32 parameter arena (named_parameters order)
        self._uv: Optional[torch.Tensor] = None         # fp32 spectral-norm arena (named_buffers order)
        self._offsets: Dict[str, int] = {}
        self._workspaces: Dict[tuple, List[_DWorkspace]] = {}
        self.grad_hook = None   # optional callable(flat_grad) after a backward that produced weight gradients
        self.__dict__["_flat_param"] = None   # see flat_parameter(); kept out of nn.Module's parameter registry

    # ---- arenas ---------------------------------------------------------------------------------------------
    def _ordered_params(self) -> List[nn.Parameter]:
        return [p for _, p in self.named_parameters()]

    @staticmethod
    def _is_arena(flat: Optional[torch.Tensor], tensors) -> bool:
        if flat is None:
            return False
        off = 0
        for t in tensors:
            if t.dtype != torch.float32 or t.data_ptr() != flat.data_ptr() + 4 * off:
                return False
            off += t.numel()
        return off == flat.numel()

    def flat_parameters(self) -> torch.Tensor:
        params = self._ordered_params()
        if not self._is_arena(self._flat, params):
            total = sum(p.numel() for p in params)
            if total != _lib.lib().resr_discriminator_param_count():
                raise RuntimeError("Discriminator: parameter count differs from the native plan")
            flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
            off = 0
            self._offsets = {}
            for name, p in self.named_parameters():
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1).float())
                p.data = flat[off:off + n].view(p.shape)
                self._offsets[name] = off
                off += n
            self._flat = flat
            self._workspaces.clear()
        return self._flat

    def flat_parameter(self) -> nn.Parameter:
        """One leaf Parameter aliasing the whole fp32 arena (as `Generator.flat_parameter`): `optim.Adam([d.flat_parameter()],
        fused=True)` and the GradScaler's unscale / inf check are single launches, and the autograd graph carries ONE parameter
        input instead of 19 (no per-tensor AccumulateGrad copies / adds: the two backward passes of a GAN step,
        train_realesrgan.py:503-516, meet in one `add_` over the arena).  After this call backward ACCUMULATES the gradient arena
        into this Parameter's `.grad` (autograd semantics) and hands out no per-tensor gradients; `zero_grad()` clears it; setting
        `requires_grad` on the module's parameters (the frozen discriminator of the generator step, :465-466) is mirrored onto it
        by `requires_grad_`.  `state_dict()` is unchanged."""
        flat = self.flat_parameters()
        fp = self.__dict__["_flat_param"]
        if fp is None:
            fp = nn.Parameter(flat, requires_grad=True)
            self.__dict__["_flat_param"] = fp
        elif fp.data_ptr() != flat.data_ptr():
            fp.data = flat
        return fp

    def flat_grad(self) -> Optional[torch.Tensor]:
        fp = self.__dict__["_flat_param"]
        return None if fp is None else fp.grad

    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none=set_to_none)
        fp = self.__dict__["_flat_param"]
        if fp is not None and fp.grad is not None:
            if set_to_none:
                fp.grad = None
            else:
                fp.grad.zero_()

    def requires_grad_(self, requires_grad: bool = True):
        fp = self.__dict__["_flat_param"]
        if fp is not None:
            fp.requires_grad_(requires_grad)
        return super().requires_grad_(requires_grad)

    def flat_uv(self) -> torch.Tensor:
        """The spectral-norm vectors (`weight_u`, `weight_v` of the eight normalised convs, named_buffers order) as views of
        one arena: the native forward updates them in place, a data-parallel broadcast moves them in one message."""
        named = list(self.named_buffers())
        if not self._is_arena(self._uv, [b for _, b in named]):
            total = sum(b.numel() for _, b in named)
            if total != _lib.lib().resr_discriminator_uv_count():
                raise RuntimeError("Discriminator: spectral-norm buffer count differs from the native plan")
            flat = torch.empty(total, dtype=torch.float32, device=named[0][1].device)
            off = 0
            for name, b in named:
                n = b.numel()
                flat[off:off + n].copy_(b.reshape(-1).float())
                mod, leaf = name.rsplit(".", 1)
                setattr(self.get_submodule(mod), leaf, flat[off:off + n].view(b.shape))
                off += n
            self._uv = flat
        return self._uv

    # ---- C-ABI plumbing ---------------------------------------------------------------------------------------
    def _workspace(self, desc, device) -> _DWorkspace:
        key = (desc.n, desc.h, desc.w, desc.training, desc.dtype)
        pool = self._workspaces.setdefault(key, [])
        for ws in pool:
            if not ws.busy and ws.buf.device == device:
                return ws
        nbytes = _lib.lib().resr_discriminator_workspace_bytes(C.byref(desc))
        if nbytes == 0:
            raise RuntimeError("Discriminator: expected [N,3,H,W] with H, W divisible by 8")
        ws = _DWorkspace(nbytes, device, desc)
        pool.append(ws)
        return ws

    def _run_forward(self, x: torch.Tensor, training: bool):
        L = _lib.lib()
        _lib.require_cuda(x, "Discriminator.forward")
        flat, uv = self.flat_parameters(), self.flat_uv()
        _lib.require_cuda(flat, "Discriminator parameters")
        n, c, h, w = x.shape
        if c != 3 or (h % 8) or (w % 8):
            raise RuntimeError("Discriminator: expected [N,3,H,W] with H, W divisible by 8")
        xc = x.detach().float().contiguous()
        desc = _lib.DiscriminatorDesc(n, h, w, self._dtype, 1 if training else 0, 1 if self.training else 0)
        ws = self._workspace(desc, xc.device)
        y = torch.empty((n, 1, h, w), dtype=torch.float32, device=xc.device)
        _lib.check(L.resr_discriminator_forward(C.byref(desc), _lib.ptr(xc), _lib.ptr(flat), _lib.ptr(uv), _lib.ptr(ws.table), ws.n_chunks,
                                                _lib.ptr(ws.buf), ws.buf.numel(), _lib.ptr(y), _lib.stream_ptr(xc)),
                   "resr_discriminator_forward")
        return y, desc, ws

    def _run_backward(self, desc, ws: _DWorkspace, gy: torch.Tensor, need_gx: bool, need_w: bool):
        L = _lib.lib()
        flat = self.flat_parameters()
        # a fresh arena per backward: autograd adds the two backward passes of a GAN step (train_realesrgan.py:503-516) itself,
        # and the per-parameter views it is handed stay consecutive in memory (one all-reduce for the data-parallel exchange)
        gflat = torch.zeros_like(flat) if need_w else None
        gx = torch.empty((desc.n, 3, desc.h, desc.w), dtype=torch.float32, device=gy.device) if need_gx else None
        _lib.check(L.resr_discriminator_backward(C.byref(desc), _lib.ptr(gy), _lib.ptr(flat), _lib.ptr(ws.buf), ws.buf.numel(),
                                                 _lib.ptr(gflat), _lib.ptr(gx), _lib.stream_ptr(gy)), "resr_discriminator_backward")
        if not need_w:
            return [None] * len(self._ordered_params()), gx
        if self.grad_hook is not None:
            self.grad_hook(gflat)
        fp = self.__dict__["_flat_param"]
        if fp is not None:                        # flat_parameter() mode: accumulate into the alias, no per-tensor gradients
            if fp.data_ptr() != flat.data_ptr():
                fp.data = flat
            if fp.grad is None:
                fp.grad = gflat
            else:
                fp.grad.add_(gflat)
            return [None], gx
        grads = []
        for name, p in self.named_parameters():
            off = self._offsets[name]
            grads.append(gflat[off:off + p.numel()].view(p.shape))
        return grads, gx

    # ---- module surface -----------------------------------------------------------------------------------------
    def _forward_impl(self, x: torch.Tensor) -> torch.Tensor:
        flat = self.flat_parameters()
        fp = self.__dict__["_flat_param"]
        if fp is not None and fp.data_ptr() != flat.data_ptr():
            fp.data = flat                                               # the arena was rebuilt (.to(), new tensors loaded): keep the alias on it
        params = [fp] if fp is not None else self._ordered_params()     # flat_parameter() mode: one graph input for all 19 tensors
        training = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _DiscFn.apply(self, training, x, *params)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._forward_impl(x)
