"""U-Net discriminator with spectral norm (reference model.py:135-203) on the gfx950 kernels.

Same constructor, forward semantics ([N,3,H,W] -> [N,1,H,W] logits, H,W divisible by 8) and state_dict
keys as the reference (`conv1.*`, `{down_block1..3,up_block1..3,conv2,conv3}.0.{weight_orig,weight_u,
weight_v}`, `conv4.*`): the parameter containers are literally `spectral_norm(nn.Conv2d(...))` objects
built in the reference's order, so the init RNG stream is identical; their Python forward is never used.

Execution plan (all through the C-ABI, include/resr.h):
  * every convolution runs on `resr_conv3x3` (MFMA implicit GEMM), 64 output channels per launch;
  * the three 4x4 / stride-2 convs are 3x3 convs over the 2x2 space-to-depth image with a sparse
    "virtual" kernel produced by `resr_pack_weights` (virtual4x4) -- no separate strided-conv kernel;
  * spectral norm: `resr_spectral_norm` (one power iteration per training-mode forward, u/v updated in
    place like torch's hook), 1/sigma folded into the packed weights through a device scalar (no sync);
  * bilinear x2, skip adds, LeakyReLU masks: `resr_bilinear_up2x`, conv epilogues, `resr_add_mask`;
  * backward: mirrored data-gradient convs with transposed packs, `resr_conv3x3_wgrad` for weights,
    `resr_fold4x4` + `resr_spectral_norm_bwd` to map gradients back to `weight_orig`.
Orchestration is Python in this round (about 60 launches forward, 150 backward).
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

SLOPE = 0.2
# name, cin (real), cout, virtual4x4, spectral norm, bias
_LAYERS = [("conv1", 3, 64, False, False, True),
           ("down_block1.0", 64, 128, True, True, False), ("down_block2.0", 128, 256, True, True, False),
           ("down_block3.0", 256, 512, True, True, False),
           ("up_block1.0", 512, 256, False, True, False), ("up_block2.0", 256, 128, False, True, False),
           ("up_block3.0", 128, 64, False, True, False),
           ("conv2.0", 64, 64, False, True, False), ("conv3.0", 64, 64, False, True, False),
           ("conv4", 64, 1, False, False, True)]


def _r32(v: int) -> int:
    return (v + 31) // 32 * 32


class _Plan:
    """Packed-weight layout + chunk tables for one precision (built once per module/device)."""

    def __init__(self, offsets: Dict[str, int]):
        self.fwd, self.bwd = {}, {}          # name -> list of (dst_off_elems, mt, cin_pad) per cout group
        chunks: List[tuple] = []
        off = 0
        for li, (name, cin, cout, virt, sn, _b) in enumerate(_LAYERS):
            cin_v = cin * 4 if virt else cin
            cin_pad, cout_pad = _r32(cin_v), _r32(cout)
            # forward: M = cout in groups of <= 64, K = cin_v
            groups = []
            for g0 in range(0, cout_pad, 64):
                mt = min(64, cout_pad - g0) // 32
                groups.append((off, mt))
                for ck in range(cin_pad // 32):
                    chunks.append((offsets[name], off, cout, cin, g0, max(0, min(64, cout - g0)), ck * 32,
                                   max(0, min(32, cin_v - ck * 32)), mt, 0, 1 if virt else 0, li if sn else -1))
                    off += 9 * mt * 1024
            self.fwd[name] = groups
            # backward-data: M = cin_v in groups of <= 64, K = cout
            groups = []
            for g0 in range(0, cin_pad, 64):
                mt = min(64, cin_pad - g0) // 32
                groups.append((off, mt))
                for ck in range(cout_pad // 32):
                    chunks.append((offsets[name], off, cout, cin, g0, max(0, min(64, cin_v - g0)), ck * 32,
                                   max(0, min(32, cout - ck * 32)), mt, 1, 1 if virt else 0, li if sn else -1))
                    off += 9 * mt * 1024
            self.bwd[name] = groups
        self.chunks, self.total_elems = chunks, off


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "Discriminator", training: bool, x: torch.Tensor, *params: torch.Tensor):
        y, saved = module._run_forward(x, training)
        ctx.module, ctx.saved = module, saved
        ctx.x_needs_grad = x.requires_grad
        ctx.param_needs = [p.requires_grad for p in params]
        return y

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        grads, gx = ctx.module._run_backward(ctx.saved, gy.contiguous().float(), ctx.x_needs_grad, any(ctx.param_needs))
        ctx.saved = None
        return (None, None, gx) + tuple(g if need else None for g, need in zip(grads, ctx.param_needs))


class Discriminator(nn.Module):
    """Reference `Discriminator()` (model.py:135-203).  Extra keyword `precision` as for `Generator`."""

    def __init__(self, precision: Optional[str] = None) -> None:
        super().__init__()
        self.precision = precision or os.environ.get("RESR_PRECISION", "fast")
        self._dtype = _precision_to_dtype(self.precision)
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
        self._flat: Optional[torch.Tensor] = None
        self._plan: Optional[_Plan] = None
        self._table: Optional[torch.Tensor] = None
        self._sigma: Optional[torch.Tensor] = None
        self.grad_hook = None

    # ---- parameters -------------------------------------------------------------------------------------
    def _wname(self, name: str) -> str:
        sn = dict((l[0], l[4]) for l in _LAYERS)[name]
        return name + (".weight_orig" if sn else ".weight")

    def _ordered_params(self) -> List[nn.Parameter]:
        return [p for _, p in self.named_parameters()]

    def _flatten(self) -> None:
        params = self._ordered_params()
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
        off = 0
        self._offsets = {}
        for (name, p) in self.named_parameters():
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = flat[off:off + n].view(p.shape)
            self._offsets[name] = off
            off += n
        self._flat, self._plan, self._table, self._sigma = flat, None, None, None

    def flat_parameters(self) -> torch.Tensor:
        ok = self._flat is not None
        if ok:
            off = 0
            for p in self._ordered_params():
                if p.data_ptr() != self._flat.data_ptr() + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            self._flatten()
        return self._flat

    def _buf(self, name: str) -> torch.Tensor:
        return dict(self.named_buffers())[name]

    # ---- packing ------------------------------------------------------------------------------------------
    def _ensure_plan(self) -> None:
        flat = self.flat_parameters()
        if self._plan is not None and self._table is not None and self._table.device == flat.device:
            return
        offs = {l[0]: self._offsets[self._wname(l[0])] for l in _LAYERS}
        self._plan = _Plan(offs)
        self._sigma = torch.ones(len(_LAYERS), 2, dtype=torch.float32, device=flat.device)
        host = (_lib.PackChunk * len(self._plan.chunks))()
        for i, (so, do, sc, scin, mo, mc, ko, kc, mt, tr, virt, li) in enumerate(self._plan.chunks):
            sp = (self._sigma.data_ptr() + (li * 2 + 1) * 4) if li >= 0 else None
            host[i] = _lib.PackChunk(so, do, sc, scin, mo, mc, ko, kc, mt, tr, 1.0, virt, sp)
        self._table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(flat.device)

    # ---- low-level launch helpers ---------------------------------------------------------------------------
    def _T(self):
        return torch.float16 if self._dtype == _lib.RESR_F16 else torch.float32

    def _new(self, n, h, w, c):
        return torch.empty((n, h, w, c), dtype=self._T(), device=self._flat.device)

    def _conv(self, packed, groups, x, cin_pad, n, h, w, cout, out, flags=0, bias=None, res0=None, mask=None, aux=None,
              s2d_in=0, s2d_out=0):
        """3x3 conv of NHWC `x` (first cin_pad channels) into NHWC `out` ([..., C_out_total]) in 64-channel groups.
        s2d_in = C: `x` is a space-to-depth image with C channels per sub-position (a 4x4 / stride-2 conv, forward);
        s2d_out = C: `out` is the gradient of one (backward-data of such a conv) -- the kernel then skips the virtual
        kernel's zero taps (16 tap-products instead of 36)."""
        L, lib = _lib, _lib.lib()
        es = x.element_size()
        cout_pad_total = _r32(cout)
        st = _lib.stream_ptr(x)
        nchw = bool(flags & L.CONV_OUT_NCHW_F32)
        if (len(groups) > 1 and self._dtype == L.RESR_F16 and cout == 64 * len(groups) and bias is None and not nchw
                and (flags & L.CONV_NO_BIAS)):
            # all 64-channel output groups of the layer in ONE launch (they would each under-fill the GPU at 32^2..128^2 pixels)
            d = L.ConvDesc(n, h, w, cin_pad, cin_pad, x.shape[-1], 0, 64, 64, out.shape[-1],
                           res0.shape[-1] if res0 is not None else 0, 0, mask.shape[-1] if mask is not None else 0,
                           self._dtype, flags, 1.0, 1.0, 1.0, 1.0, SLOPE)
            d.cout_groups = len(groups)
            d.s2d_in_channels, d.s2d_out_channels = s2d_in, s2d_out
            L.check(lib.resr_conv3x3(C.byref(d), L.ptr(x), None, C.c_void_p(packed.data_ptr() + groups[0][0] * es), None,
                                     L.ptr(res0), None, L.ptr(mask), L.ptr(out), L.ptr(aux), st), "resr_conv3x3")
            return
        for gi, (off, mt) in enumerate(groups):
            g0 = gi * 64
            co = max(0, min(mt * 32, cout - g0))
            if co == 0:
                continue
            d = L.ConvDesc(n, h, w, cin_pad, cin_pad, x.shape[-1], 0, co, mt * 32, 0 if nchw else out.shape[-1],
                           res0.shape[-1] if res0 is not None else 0, 0, mask.shape[-1] if mask is not None else 0,
                           self._dtype, flags, 1.0, 1.0, 1.0, 1.0, SLOPE)
            d.s2d_in_channels = s2d_in
            sh = lambda t: None if t is None else C.c_void_p(t.data_ptr() + g0 * t.element_size())
            L.check(lib.resr_conv3x3(C.byref(d), L.ptr(x), None, C.c_void_p(packed.data_ptr() + off * es),
                                     None if bias is None else C.c_void_p(bias.data_ptr() + g0 * 4),
                                     sh(res0), None, sh(mask), L.ptr(out) if nchw else sh(out), sh(aux), st), "resr_conv3x3")
        del cout_pad_total

    def _wgrad(self, x, cin_pad, cin_real, g, cout, n, h, w, dw, db=None):
        """dW[cout][cin_real][3][3] (+db) = wgrad(X = first cin_pad channels of x, G = first cout channels of g)."""
        L, lib = _lib, _lib.lib()
        chunks = cin_pad // 32
        step = 64 if chunks * 2 <= 80 else 32          # wgrad.hip kMaxJobs products per launch
        th = 8 if self._dtype == L.RESR_F16 else 4
        tiles = ((w + 31) // 32) * ((h + th - 1) // th) * n
        st = _lib.stream_ptr(x)
        for g0 in range(0, _r32(cout), step):
            co = max(0, min(step, cout - g0))
            if co == 0:
                continue
            cp = _r32(co)
            jobs = chunks * (cp // 32)
            if self._dtype == L.RESR_F16:               # quad kernel: jobs/4 workgroups per split, one per CU (generator.hip splits_for)
                splits = 512 // ((jobs + 3) // 4)
                splits = min(256, splits & ~7 if splits >= 16 else splits)
            else:
                splits = min(128, 768 // jobs)
            splits = max(1, min(splits, max(1, tiles // 2)))
            d = L.WgradDesc(n, h, w, cin_pad, cin_pad, x.shape[-1], 0, cin_real, co, cp, g.shape[-1], self._dtype, 0, splits, 1.0)
            nbytes = lib.resr_wgrad_partial_bytes(C.byref(d))
            part = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
            L.check(lib.resr_conv3x3_wgrad(C.byref(d), L.ptr(x), None, C.c_void_p(g.data_ptr() + g0 * g.element_size()),
                                           L.ptr(part), C.c_void_p(dw.data_ptr() + g0 * cin_real * 9 * 4),
                                           None if db is None else C.c_void_p(db.data_ptr() + g0 * 4), st), "resr_conv3x3_wgrad")

    # ---- forward ----------------------------------------------------------------------------------------------
    def _run_forward(self, x: torch.Tensor, training: bool):
        L, lib = _lib, _lib.lib()
        _lib.require_cuda(x, "Discriminator.forward")
        flat = self.flat_parameters()
        _lib.require_cuda(flat, "Discriminator parameters")
        self._ensure_plan()
        n, c, S, S2 = x.shape
        if c != 3 or (S % 8) or (S2 % 8):
            raise RuntimeError("Discriminator: expected [N,3,H,W] with H, W divisible by 8")
        st = _lib.stream_ptr(x)
        T, dt = self._T(), self._dtype
        sn_training = self.training
        # spectral norm: power iteration (training mode), sigma per layer on the device
        tmp = torch.empty(512 + 16 * 4608 + 8, dtype=torch.float32, device=flat.device)   # rows + ceil(rows/32) * cols of the largest layer
        for li, (name, cin, cout, virt, sn, _b) in enumerate(_LAYERS):
            if not sn:
                continue
            wt = dict(self.named_parameters())[name + ".weight_orig"]
            u, v = self._buf(name + ".weight_u"), self._buf(name + ".weight_v")
            L.check(lib.resr_spectral_norm(L.ptr(wt), L.ptr(u), L.ptr(v), cout, wt.numel() // cout, 1 if sn_training else 0,
                                           1e-12, C.c_void_p(self._sigma.data_ptr() + li * 8), L.ptr(tmp), st), "resr_spectral_norm")
        es = 2 if dt == L.RESR_F16 else 4
        packed = torch.zeros(self._plan.total_elems * es + 16384, dtype=torch.uint8, device=flat.device)
        L.check(lib.resr_pack_weights(L.ptr(self._table), len(self._plan.chunks), L.ptr(flat), L.ptr(packed), dt, st), "resr_pack_weights")
        P = self._plan.fwd
        bias1 = self.conv1.bias
        bias4 = self.conv4.bias
        xc = x.detach().float().contiguous()
        x_in = self._new(n, S, S2, 32)
        L.check(lib.resr_nchw_to_nhwc(L.ptr(xc), L.ptr(x_in), n, 3, S, S2, 1, 32, dt, None, st))
        H1, W1, H2, W2_, H3, W3 = S // 2, S2 // 2, S // 4, S2 // 4, S // 8, S2 // 8
        out1 = self._new(n, S, S2, 64)
        self._conv(packed, P["conv1"], x_in, 32, n, S, S2, 64, out1, 0, bias=bias1)

        def s2d(src, h, w, cch):
            dst = self._new(n, h // 2, w // 2, 4 * cch)
            L.check(lib.resr_space_to_depth(L.ptr(src), L.ptr(dst), n, h, w, cch, dt, 0, st))
            return dst

        def up(src, h, w, cch):
            dst = self._new(n, 2 * h, 2 * w, cch)
            L.check(lib.resr_bilinear_up2x(L.ptr(src), L.ptr(dst), n, h, w, cch, dt, 0, st))
            return dst

        NB = L.CONV_NO_BIAS
        s1 = s2d(out1, S, S2, 64)
        d1 = self._new(n, H1, W1, 128)
        self._conv(packed, P["down_block1.0"], s1, 256, n, H1, W1, 128, d1, L.CONV_LRELU | NB, s2d_in=64)
        s2 = s2d(d1, H1, W1, 128)
        d2 = self._new(n, H2, W2_, 256)
        self._conv(packed, P["down_block2.0"], s2, 512, n, H2, W2_, 256, d2, L.CONV_LRELU | NB, s2d_in=128)
        s3 = s2d(d2, H2, W2_, 256)
        d3 = self._new(n, H3, W3, 512)
        self._conv(packed, P["down_block3.0"], s3, 1024, n, H3, W3, 512, d3, L.CONV_LRELU | NB, s2d_in=256)
        FL = L.CONV_LRELU | NB | (L.CONV_AUX_BEFORE_RES if training else 0)
        b1 = up(d3, H3, W3, 512)
        u1, a1 = self._new(n, H2, W2_, 256), (self._new(n, H2, W2_, 256) if training else None)
        self._conv(packed, P["up_block1.0"], b1, 512, n, H2, W2_, 256, u1, FL, res0=d2, aux=a1)
        b2 = up(u1, H2, W2_, 256)
        u2, a2 = self._new(n, H1, W1, 128), (self._new(n, H1, W1, 128) if training else None)
        self._conv(packed, P["up_block2.0"], b2, 256, n, H1, W1, 128, u2, FL, res0=d1, aux=a2)
        b3 = up(u2, H1, W1, 128)
        u3, a3 = self._new(n, S, S2, 64), (self._new(n, S, S2, 64) if training else None)
        self._conv(packed, P["up_block3.0"], b3, 128, n, S, S2, 64, u3, FL, res0=out1, aux=a3)
        c2 = self._new(n, S, S2, 64)
        self._conv(packed, P["conv2.0"], u3, 64, n, S, S2, 64, c2, L.CONV_LRELU | NB)
        c3 = self._new(n, S, S2, 64)
        self._conv(packed, P["conv3.0"], c2, 64, n, S, S2, 64, c3, L.CONV_LRELU | NB)
        y = torch.empty((n, 1, S, S2), dtype=torch.float32, device=x.device)
        self._conv(packed, P["conv4"], c3, 64, n, S, S2, 1, y, L.CONV_OUT_NCHW_F32, bias=bias4)
        saved = None
        if training:
            uv = {name: (self._buf(name + ".weight_u").clone(), self._buf(name + ".weight_v").clone())
                  for (name, _c, _o, _v, sn, _b) in _LAYERS if sn}
            saved = dict(n=n, S=S, S2=S2, packed=packed, sigma=self._sigma.clone(), uv=uv, x_in=x_in, out1=out1, s1=s1, d1=d1,
                         s2=s2, d2=d2, s3=s3, d3=d3, b1=b1, a1=a1, b2=b2, a2=a2, b3=b3, a3=a3, u3=u3, c2=c2, c3=c3)
        return y, saved

    # ---- backward ---------------------------------------------------------------------------------------------
    def _run_backward(self, s, gy: torch.Tensor, need_gx: bool, need_w: bool):
        L, lib = _lib, _lib.lib()
        st = _lib.stream_ptr(gy)
        dt = self._dtype
        n, S, S2, packed = s["n"], s["S"], s["S2"], s["packed"]
        H1, W1, H2, W2_, H3, W3 = S // 2, S2 // 2, S // 4, S2 // 4, S // 8, S2 // 8
        B = self._plan.bwd
        flat = self._flat
        gflat = torch.zeros_like(flat) if need_w else None
        named = dict(self.named_parameters())

        def gview(pname):
            off = self._offsets[pname]
            return gflat[off:off + named[pname].numel()]

        tmp1 = torch.zeros(1, dtype=torch.float32, device=flat.device)

        def wgrad_layer(name, x, cin_pad, cin_real_v, g, cout, h, w, li, virt_c=0, bias=False):
            if not need_w:
                return
            sn = li is not None
            wname = name + (".weight_orig" if sn else ".weight")
            dst = gview(wname)
            raw = dst if (not sn and not virt_c) else torch.empty(cout * cin_real_v * 9, dtype=torch.float32, device=flat.device)
            self._wgrad(x, cin_pad, cin_real_v, g, cout, n, h, w, raw, gview(name + ".bias") if bias else None)
            if virt_c:
                folded = torch.empty(cout * virt_c * 16, dtype=torch.float32, device=flat.device)
                L.check(lib.resr_fold4x4(L.ptr(raw), L.ptr(folded), cout, virt_c, st))
                raw = folded
            if sn:
                u, v = s["uv"][name]
                L.check(lib.resr_spectral_norm_bwd(L.ptr(raw), L.ptr(named[wname]), L.ptr(u), L.ptr(v),
                                                   C.c_void_p(s["sigma"].data_ptr() + li * 8), L.ptr(dst), cout,
                                                   raw.numel() // cout, 0, L.ptr(tmp1), st), "resr_spectral_norm_bwd")

        def add_mask(a, b, mask):
            out = torch.empty_like(a)
            L.check(lib.resr_add_mask(L.ptr(a), L.ptr(b), L.ptr(mask), L.ptr(out), a.numel(), dt, SLOPE, st))
            return out

        def d2s(src, h, w, cch):     # [n,h/2,w/2,4c] -> [n,h,w,c]
            dst = self._new(n, h, w, cch)
            L.check(lib.resr_space_to_depth(L.ptr(src), L.ptr(dst), n, h, w, cch, dt, 1, st))
            return dst

        def up_bwd(g, h, w, cch):    # g: [n,2h,2w,c] -> [n,h,w,c]
            dst = self._new(n, h, w, cch)
            L.check(lib.resr_bilinear_up2x(L.ptr(g), L.ptr(dst), n, h, w, cch, dt, 1, st))
            return dst

        NB = L.CONV_NO_BIAS
        LI = {l[0]: i for i, l in enumerate(_LAYERS)}
        g4 = self._new(n, S, S2, 32)
        L.check(lib.resr_nchw_to_nhwc(L.ptr(gy), L.ptr(g4), n, 1, S, S2, 1, 32, dt, None, st))
        wgrad_layer("conv4", s["c3"], 64, 64, g4, 1, S, S2, None, bias=True)
        G8 = self._new(n, S, S2, 64)
        self._conv(packed, B["conv4"], g4, 32, n, S, S2, 64, G8, NB | L.CONV_MASK, mask=s["c3"])
        wgrad_layer("conv3.0", s["c2"], 64, 64, G8, 64, S, S2, LI["conv3.0"])
        G7 = self._new(n, S, S2, 64)
        self._conv(packed, B["conv3.0"], G8, 64, n, S, S2, 64, G7, NB | L.CONV_MASK, mask=s["c2"])
        wgrad_layer("conv2.0", s["u3"], 64, 64, G7, 64, S, S2, LI["conv2.0"])
        G6, g_u3 = self._new(n, S, S2, 64), self._new(n, S, S2, 64)
        self._conv(packed, B["conv2.0"], G7, 64, n, S, S2, 64, G6, NB | L.CONV_MASK | L.CONV_AUX_BEFORE_MASK, mask=s["a3"], aux=g_u3)
        wgrad_layer("up_block3.0", s["b3"], 128, 128, G6, 64, S, S2, LI["up_block3.0"])
        g_b3 = self._new(n, S, S2, 128)
        self._conv(packed, B["up_block3.0"], G6, 64, n, S, S2, 128, g_b3, NB)
        g_u2 = up_bwd(g_b3, H1, W1, 128)
        G5 = add_mask(g_u2, None, s["a2"])
        wgrad_layer("up_block2.0", s["b2"], 256, 256, G5, 128, H1, W1, LI["up_block2.0"])
        g_b2 = self._new(n, H1, W1, 256)
        self._conv(packed, B["up_block2.0"], G5, 128, n, H1, W1, 256, g_b2, NB)
        g_u1 = up_bwd(g_b2, H2, W2_, 256)
        G4 = add_mask(g_u1, None, s["a1"])
        wgrad_layer("up_block1.0", s["b1"], 512, 512, G4, 256, H2, W2_, LI["up_block1.0"])
        g_b1 = self._new(n, H2, W2_, 512)
        self._conv(packed, B["up_block1.0"], G4, 256, n, H2, W2_, 512, g_b1, NB)
        g_d3 = up_bwd(g_b1, H3, W3, 512)
        G3 = add_mask(g_d3, None, s["d3"])
        wgrad_layer("down_block3.0", s["s3"], 1024, 1024, G3, 512, H3, W3, LI["down_block3.0"], virt_c=256)
        g_s3 = self._new(n, H3, W3, 1024)
        self._conv(packed, B["down_block3.0"], G3, 512, n, H3, W3, 1024, g_s3, NB, s2d_out=256)
        G2 = add_mask(d2s(g_s3, H2, W2_, 256), g_u1, s["d2"])
        wgrad_layer("down_block2.0", s["s2"], 512, 512, G2, 256, H2, W2_, LI["down_block2.0"], virt_c=128)
        g_s2 = self._new(n, H2, W2_, 512)
        self._conv(packed, B["down_block2.0"], G2, 256, n, H2, W2_, 512, g_s2, NB, s2d_out=128)
        G1 = add_mask(d2s(g_s2, H1, W1, 128), g_u2, s["d1"])
        wgrad_layer("down_block1.0", s["s1"], 256, 256, G1, 128, H1, W1, LI["down_block1.0"], virt_c=64)
        g_s1 = self._new(n, H1, W1, 256)
        self._conv(packed, B["down_block1.0"], G1, 128, n, H1, W1, 256, g_s1, NB, s2d_out=64)
        G0 = add_mask(d2s(g_s1, S, S2, 64), g_u3, None)
        wgrad_layer("conv1", s["x_in"], 32, 3, G0, 64, S, S2, None, bias=True)
        gx = None
        if need_gx:
            gxin = self._new(n, S, S2, 32)
            self._conv(packed, B["conv1"], G0, 64, n, S, S2, 32, gxin, NB)
            gx = torch.empty((n, 3, S, S2), dtype=torch.float32, device=gy.device)
            L.check(lib.resr_nhwc_to_nchw(L.ptr(gxin), L.ptr(gx), n, 3, S, S2, 1, 32, dt, st))
        grads = []
        if need_w:
            if self.grad_hook is not None:
                self.grad_hook(gflat)
            for name, p in self.named_parameters():
                off = self._offsets[name]
                grads.append(gflat[off:off + p.numel()].view(p.shape))
        else:
            grads = [None] * len(self._ordered_params())
        return grads, gx

    # ---- module surface ------------------------------------------------------------------------------------------
    def _forward_impl(self, x: torch.Tensor) -> torch.Tensor:
        self.flat_parameters()
        params = self._ordered_params()
        training = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _DiscFn.apply(self, training, x, *params)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._forward_impl(x)
