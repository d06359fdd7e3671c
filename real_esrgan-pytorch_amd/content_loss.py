"""VGG19 perceptual loss (reference model.py:278-335) on the MFMA conv kernel.

The reference's GAN step wraps the five losses in `torch.Tensor(...)` (train_realesrgan.py:477-478), so the perceptual
term is detached there: it is logged but never back-propagated.  `detached=True` (default) reproduces that quirk --
forward only, detached scalars.  `detached=False` is the graph the reference WROTE (model.py:311-335 with the weights of
config.py:137): the five L1 terms back-propagate into `sr_tensor` through VGG19 -- ReLU masks, 2x2 max-pool argmax and
backward-data convolutions on the same kernel; the VGG weights are frozen (model.py:306-308), so there are no weight
gradients.  16 x conv3x3 + ReLU, 4 x max-pool, ImageNet normalisation, L1 per tapped node.

Weights: torchvision's pretrained VGG19 cannot be downloaded here, so the convs start from torchvision's VGG
*initialisation* (kaiming_normal fan_out / zero bias); `load_state_dict` accepts a torchvision `vgg19()` state
dict (`features.N.weight|bias`) when one is available.  Numerics vs the real torchvision graph are therefore
unpinned; in particular torchvision's ReLU(inplace=True) makes every tapped conv output except the
last one alias its ReLU-ed value -- reproduced here under `inplace_relu_aliasing=True` (default); checked against a torch.fx
extractor over a plain-torch clone of the module structure (tests/test_oracle_vgg_extractor.py), not against torchvision itself.

precision: "fast" (f16), "exact16" (hi/lo f16 pairs, fp32-class) or "strict" (f32), as for `Generator`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .model import _precision_to_dtype

_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512]


def _r32(v):
    return (v + 31) // 32 * 32


class _Act:
    """One NHWC activation of the native graph: `t` holds the hi tensor and, for exact16, the lo tensor right behind it
    ([2, n, h, w, c]); `lo` = the hi -> lo element offset (0 outside exact16)."""

    def __init__(self, n, h, w, c, dtype, device):
        L = _lib
        self.n, self.h, self.w, self.c = n, h, w, c
        T = torch.float32 if dtype == L.RESR_F32 else torch.float16
        pairs = 2 if dtype == L.RESR_F16X2 else 1
        self.t = torch.empty((pairs, n, h, w, c), dtype=T, device=device)
        self.lo = n * h * w * c if pairs == 2 else 0
        self.es = 4 if dtype == L.RESR_F32 else 2

    def ptr(self, chan: int = 0):
        return C.c_void_p(self.t.data_ptr() + chan * self.es)

    def value(self, n: Optional[int] = None) -> torch.Tensor:
        """fp32 [n, h, w, c] (hi + lo * 2^-12)."""
        hi = self.t[0, :n].float()
        return hi if self.t.shape[0] == 1 else hi + self.t[1, :n].float() * (1.0 / 4096.0)

    def assign(self, v: torch.Tensor) -> None:
        """Store an fp32 [n, h, w, c] tensor (split into hi / lo for exact16)."""
        if self.t.shape[0] == 1:
            self.t[0].copy_(v)
            return
        hi = v.to(torch.float16)
        self.t[0].copy_(hi)
        self.t[1].copy_((v - hi.float()) * 4096.0)


class _FeatureFn(torch.autograd.Function):
    """(sr, hr) -> the tapped features of both as fp32 NHWC tensors; backward: the sr half through the native VGG backward."""

    @staticmethod
    def forward(ctx, module: "ContentLoss", sr: torch.Tensor, hr: torch.Tensor):
        b = sr.shape[0]
        feats, saved = module._features(torch.cat([sr.detach(), hr.detach()], 0), keep=True)
        ctx.module, ctx.saved, ctx.b = module, saved, b
        outs = []
        for k in module.feature_model_extractor_nodes:
            outs.append(feats[k].value()[:b].contiguous())
        for k in module.feature_model_extractor_nodes:
            outs.append(feats[k].value()[b:].contiguous())
        ctx.mark_non_differentiable(*outs[len(outs) // 2:])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        k = len(ctx.module.feature_model_extractor_nodes)
        g = {name: grads[i] for i, name in enumerate(ctx.module.feature_model_extractor_nodes) if grads[i] is not None}
        return None, ctx.module._backward(ctx.saved, g, ctx.b), None


class ContentLoss(nn.Module):
    def __init__(self, feature_model_extractor_nodes: list, feature_model_normalize_mean: list,
                 feature_model_normalize_std: list, precision: Optional[str] = None,
                 inplace_relu_aliasing: bool = True, detached: bool = True) -> None:
        super().__init__()
        self.feature_model_extractor_nodes = feature_model_extractor_nodes
        self.precision = precision or os.environ.get("RESR_PRECISION", "fast")
        self._dtype = _precision_to_dtype(self.precision)
        self.aliasing = inplace_relu_aliasing
        self.detached = detached
        self.features = nn.Module()
        self.layers: List[tuple] = []            # ("conv", idx, cin, cout) | ("pool", idx)
        idx, cin = 0, 3
        for v in _CFG:
            if v == "M":
                self.layers.append(("pool", idx))
                idx += 1
            else:
                conv = nn.Conv2d(cin, v, 3, padding=1)
                nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")   # torchvision VGG._initialize
                nn.init.constant_(conv.bias, 0)
                self.features.add_module(str(idx), conv)
                self.layers.append(("conv", idx, cin, v))
                idx += 2                          # conv + ReLU
                cin = v
        self.register_buffer("mean", torch.tensor(feature_model_normalize_mean).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor(feature_model_normalize_std).view(1, 3, 1, 1))
        for p in self.parameters():               # reference model.py:307-309
            p.requires_grad = False
        self._packed = None

    # ---- weights ---------------------------------------------------------------------------------------------
    def _pack(self, device):
        """Pack all 16 convs once (frozen weights): the forward form per conv -- cout groups of 64, K chunks of 32 -- and the
        backward-data form (M = cin groups of 64, K = cout chunks, taps flipped)."""
        if self._packed is not None and self._packed[0].device == device:
            return
        L = _lib
        flat = torch.cat([getattr(self.features, str(l[1])).weight.detach().float().reshape(-1) for l in self.layers if l[0] == "conv"]).to(device)
        chunks, fwd, bwd, off, src = [], {}, {}, 0, 0
        for l in self.layers:
            if l[0] != "conv":
                continue
            _, idx, cin, cout = l
            for table, m_real, k_real, tr in ((fwd, cout, cin, 0), (bwd, cin, cout, 1)):
                gl = []
                for g0 in range(0, _r32(m_real), 64):
                    mt = min(64, _r32(m_real) - g0) // 32
                    gl.append((off, mt))
                    for ck in range(_r32(k_real) // 32):
                        chunks.append(L.PackChunk(src, off, cout, cin, g0, max(0, min(64, m_real - g0)), ck * 32,
                                                  max(0, min(32, k_real - ck * 32)), mt, tr, 1.0, 0, None))
                        off += 9 * mt * 1024
                table[idx] = gl
            src += cout * cin * 9
        host = (L.PackChunk * len(chunks))(*chunks)
        table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(device)
        self._wes = {L.RESR_F16: 2, L.RESR_F32: 4, L.RESR_F16X2: 6}[self._dtype]
        packed = torch.zeros(off * self._wes + 16384, dtype=torch.uint8, device=device)
        L.check(L.lib().resr_pack_weights(L.ptr(table), len(chunks), L.ptr(flat), L.ptr(packed), self._dtype, L.stream_ptr(flat)),
                "resr_pack_weights")
        self._packed = (packed, fwd, bwd)

    def load_state_dict(self, state_dict, strict: bool = True):
        self._packed = None
        sd = {("features." + k[len("features."):]) if k.startswith("features.") else k: v for k, v in state_dict.items()}
        sd = {k: v for k, v in sd.items() if k.startswith("features.") and int(k.split(".")[1]) <= 34}
        sd.setdefault("mean", self.mean)
        sd.setdefault("std", self.std)
        return super().load_state_dict(sd, strict=strict)

    # ---- native passes ---------------------------------------------------------------------------------------
    def _conv(self, x: _Act, groups, k_real: int, m_real: int, out: _Act, flags: int, bias, aux: Optional[_Act], mask: Optional[_Act],
              n: int, st) -> None:
        """One 3x3 convolution as launches of 64 output channels: x (first r32(k_real) channels) -> out (m_real channels)."""
        L, lib = _lib, _lib.lib()
        packed = self._packed[0]
        # All 64-channel output groups of a 128..512-channel layer as ONE launch (ResrConvDesc.cout_groups: tile index = group x
        # spatial tiles, per-group packed weights, biases and channel offsets).  One launch per group left the deep layers --
        # 512 channels at 32^2 / 16^2 pixels: 128 / 32 tiles per launch -- on an eighth of the 256 CUs.
        if (len(groups) > 1 and len(groups) <= 8 and self._dtype != L.RESR_F32 and m_real == 64 * len(groups)
                and all(mt == 2 for _, mt in groups) and os.environ.get("RESR_VGG_PER_GROUP") != "1"):
            d = L.ConvDesc(n, x.h, x.w, _r32(k_real), _r32(k_real), x.c, 0, 64, 64, out.c,
                           0, 0, 0 if mask is None else mask.c, self._dtype, flags, 1.0, 1.0, 1.0, 1.0, 0.0)
            d.in0_lo_offset, d.out_lo_offset = x.lo, out.lo
            d.cout_groups = len(groups)
            L.check(lib.resr_conv3x3(C.byref(d), x.ptr(), None, C.c_void_p(packed.data_ptr() + groups[0][0] * self._wes),
                                     None if bias is None else L.ptr(bias), None, None,
                                     None if mask is None else mask.ptr(), out.ptr(), None if aux is None else aux.ptr(), st),
                    "resr_conv3x3")
            return
        for gi, (off, mt) in enumerate(groups):
            g0 = gi * 64
            d = L.ConvDesc(n, x.h, x.w, _r32(k_real), _r32(k_real), x.c, 0, min(mt * 32, m_real - g0), mt * 32, out.c,
                           0, 0, 0 if mask is None else mask.c, self._dtype, flags, 1.0, 1.0, 1.0, 1.0, 0.0)
            d.in0_lo_offset, d.out_lo_offset = x.lo, out.lo
            L.check(lib.resr_conv3x3(C.byref(d), x.ptr(), None, C.c_void_p(packed.data_ptr() + off * self._wes),
                                     None if bias is None else C.c_void_p(bias.data_ptr() + g0 * 4), None, None,
                                     None if mask is None else mask.ptr(g0), out.ptr(g0), None if aux is None else aux.ptr(g0), st),
                    "resr_conv3x3")

    def _features(self, x: torch.Tensor, keep: bool = False):
        """The tapped nodes of vgg19().features on x [n,3,h,w] as `_Act`s; keep=True also returns what the backward pass
        needs (every ReLU output, the pool argmax bytes)."""
        L, lib = _lib, _lib.lib()
        _lib.require_cuda(x, "ContentLoss")
        self._pack(x.device)
        _, fwd, _ = self._packed
        st = L.stream_ptr(x)
        xn = ((x.float() - self.mean) / self.std).contiguous()              # transforms.Normalize, model.py:317-318
        n, _, h, w = xn.shape
        cur = _Act(n, h, w, 32, self._dtype, x.device)
        L.check(lib.resr_nchw_to_nhwc(L.ptr(xn), cur.ptr(), n, 3, h, w, 1, 32, self._dtype, None, st))
        out: Dict[str, _Act] = {}
        saved = []                                   # per executed layer: ("conv", idx, cin, cout, input, relu_out) | ("pool", arg, input dims)
        wanted = {int(k.split(".")[1]) for k in self.feature_model_extractor_nodes}
        last = max(wanted)
        for l in self.layers:
            if l[0] == "pool":
                nxt = _Act(n, cur.h // 2, cur.w // 2, cur.c, self._dtype, x.device)
                arg = torch.empty((n, nxt.h, nxt.w, cur.c), dtype=torch.uint8, device=x.device) if keep else None
                L.check(lib.resr_maxpool2x2_arg(cur.ptr(), nxt.ptr(), L.ptr(arg), n, nxt.h, nxt.w, cur.c, self._dtype, st), "resr_maxpool2x2_arg")
                saved.append(("pool", arg, cur.h, cur.w, cur.c))
                cur = nxt
                continue
            _, idx, cin, cout = l
            bias = getattr(self.features, str(idx)).bias
            tap_pre = idx in wanted and (not self.aliasing or idx == last)   # the value the extractor hands back
            relu_out = _Act(n, cur.h, cur.w, cout, self._dtype, x.device)
            pre = _Act(n, cur.h, cur.w, cout, self._dtype, x.device) if tap_pre else None
            flags = L.CONV_LRELU | (L.CONV_AUX_BEFORE_MASK if tap_pre else 0)   # slope 0 => ReLU; aux = pre-activation
            self._conv(cur, fwd[idx], cin, cout, relu_out, flags, bias, pre, None, n, st)
            if idx in wanted:
                out[f"features.{idx}"] = pre if tap_pre else relu_out
            saved.append(("conv", idx, cin, cout, cur, relu_out, tap_pre))
            cur = relu_out
            if idx == last:
                break
        return out, saved

    def _backward(self, saved, grads: Dict[str, torch.Tensor], b: int) -> torch.Tensor:
        """d(sum of the tapped-feature cotangents) / d(sr): the first `b` images of the saved batch, layers in reverse."""
        L, lib = _lib, _lib.lib()
        _, _, bwd = self._packed
        dev = self.mean.device
        st = L.stream_ptr(self.mean)
        g: Optional[_Act] = None                     # gradient wrt the current layer's OUTPUT (post-ReLU / pooled)
        for rec in reversed(saved):
            if rec[0] == "pool":
                _, arg, ih, iw, c = rec
                gin = _Act(b, ih, iw, c, self._dtype, dev)
                L.check(lib.resr_maxpool2x2_bwd(g.ptr(), L.ptr(arg), gin.ptr(), b, ih // 2, iw // 2, c, self._dtype, st), "resr_maxpool2x2_bwd")
                g = gin
                continue
            _, idx, cin, cout, x_in, relu_out, tap_pre = rec
            tap = grads.get(f"features.{idx}")
            tap_act = None
            if tap is not None:
                tap_act = _Act(b, relu_out.h, relu_out.w, cout, self._dtype, dev)
                tap_act.assign(tap.float())
            count = b * relu_out.h * relu_out.w * cout
            mask_view = _view(relu_out, b)

            def add_mask(a, bb, mask):
                o = _Act(b, relu_out.h, relu_out.w, cout, self._dtype, dev)
                L.check(lib.resr_add_mask(a.ptr(), None if bb is None else bb.ptr(), None if mask is None else mask.ptr(), o.ptr(), count,
                                          self._dtype, 0.0, st), "resr_add_mask")
                return o
            # gradient wrt the pre-activation: ReLU'(pre) = (relu_out > 0); a tap of the pre-activation joins behind the mask, a
            # tap of the (aliased) post-ReLU value in front of it
            if g is None:
                g_pre = tap_act if tap_pre else add_mask(tap_act, None, mask_view)
            elif tap_act is None:
                g_pre = add_mask(g, None, mask_view)
            elif tap_pre:
                g_pre = add_mask(add_mask(g, None, mask_view), tap_act, None)
            else:
                g_pre = add_mask(g, tap_act, mask_view)
            if getattr(self, "_debug_grads", None) is not None:      # tools/diag_content_loss.py: gradient wrt every pre-activation
                self._debug_grads[idx] = g_pre.value().permute(0, 3, 1, 2).cpu()
            gin = _Act(b, x_in.h, x_in.w, x_in.c, self._dtype, dev)
            self._conv(g_pre, bwd[idx], cout, _r32(cin), gin, L.CONV_NO_BIAS, None, None, None, b, st)
            g = gin
        gx = torch.empty((b, 3, g.h, g.w), dtype=torch.float32, device=dev)
        L.check(lib.resr_nhwc_to_nchw(g.ptr(), L.ptr(gx), b, 3, g.h, g.w, 1, 32, self._dtype, st), "resr_nhwc_to_nchw")
        return gx / self.std

    def _l1_halves(self, act: _Act, b: int) -> torch.Tensor:
        """F.l1_loss(first b images, last b images) of one tapped activation, straight from the 16-bit (pair) tensor: one pass,
        deterministic partial sums (resr_l1_partial) instead of fp32 copies + sub + abs + mean."""
        L = _lib
        count = b * act.h * act.w * act.c
        nblocks = 1024
        partial = torch.empty(nblocks, dtype=torch.float32, device=act.t.device)
        second = C.c_void_p(act.t.data_ptr() + count * act.es)
        L.check(L.lib().resr_l1_partial(act.ptr(), second, count, self._dtype, act.lo, L.ptr(partial), nblocks, L.stream_ptr(act.t)),
                "resr_l1_partial")
        return partial.sum() / count

    def _l1_all(self, feats: Dict[str, _Act], nodes, b: int) -> torch.Tensor:
        """All tapped nodes' F.l1_loss(first b images, last b images) as ONE device tensor [len(nodes)]: a partial-sum launch per
        node (resr_l1_partial) into the rows of one buffer, then one launch that reduces the rows and divides (fixed order:
        deterministic) -- instead of a sum and a division per node."""
        L = _lib
        nblocks = 1024
        dev = feats[nodes[0]].t.device
        partial = torch.empty((len(nodes), nblocks), dtype=torch.float32, device=dev)
        coef = (C.c_float * len(nodes))()
        for i, k in enumerate(nodes):
            act = feats[k]
            count = b * act.h * act.w * act.c
            second = C.c_void_p(act.t.data_ptr() + count * act.es)
            L.check(L.lib().resr_l1_partial(act.ptr(), second, count, self._dtype, act.lo, C.c_void_p(partial.data_ptr() + i * nblocks * 4),
                                            nblocks, L.stream_ptr(act.t)), "resr_l1_partial")
            coef[i] = 1.0 / count
        out = torch.empty(len(nodes) + 1, dtype=torch.float32, device=dev)
        L.check(L.lib().resr_weighted_row_sums(L.ptr(partial), len(nodes), nblocks, coef, L.ptr(out), L.stream_ptr(partial)),
                "resr_weighted_row_sums")
        return out[:len(nodes)]

    # ---- module surface ----------------------------------------------------------------------------------------
    def forward(self, sr_tensor: torch.Tensor, hr_tensor: torch.Tensor):
        b = sr_tensor.shape[0]
        nodes = self.feature_model_extractor_nodes
        self.last_losses = None
        if self.detached or not (torch.is_grad_enabled() and sr_tensor.requires_grad):
            with torch.no_grad():
                feats, _ = self._features(torch.cat([sr_tensor.detach(), hr_tensor.detach()], 0))
                if len(nodes) <= 8:
                    self.last_losses = self._l1_all(feats, nodes, b)      # the values below are views of this tensor
                    return tuple(self.last_losses[i] for i in range(len(nodes)))
                return tuple(self._l1_halves(feats[k], b) for k in nodes)
        outs = _FeatureFn.apply(self, sr_tensor, hr_tensor)
        return tuple(F.l1_loss(outs[i], outs[len(nodes) + i]) for i in range(len(nodes)))


def _view(act: _Act, n: int) -> _Act:
    """The first n images of `act` as an `_Act` over the same memory (pairs keep their hi -> lo offset)."""
    v = _Act.__new__(_Act)
    v.n, v.h, v.w, v.c, v.t, v.lo, v.es = n, act.h, act.w, act.c, act.t, act.lo, act.es
    return v
