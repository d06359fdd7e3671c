"""VGG19 perceptual loss (reference model.py:278-335) on the MFMA conv kernel -- forward only.

The reference's GAN step wraps the five losses in `torch.Tensor(...)` (train_realesrgan.py:477-478), so the
perceptual term is detached: it is logged but never back-propagated.  This module therefore provides the forward
(16 x conv3x3 + ReLU, 4 x max-pool, ImageNet normalisation, L1 per tapped node) and returns detached scalars.

Weights: torchvision's pretrained VGG19 cannot be downloaded here, so the convs start from torchvision's VGG
*initialisation* (kaiming_normal fan_out / zero bias); `load_state_dict` accepts a torchvision `vgg19()` state
dict (`features.N.weight|bias`) when one is available.  Numerics vs the real torchvision graph are therefore
unpinned; in particular torchvision's ReLU(inplace=True) makes every tapped conv output except the
last one alias its ReLU-ed value -- reproduced here under `inplace_relu_aliasing=True` (default); checked against a torch.fx
extractor over a plain-torch clone of the module structure (tests/test_oracle_vgg_extractor.py), not against torchvision itself.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .model import _precision_to_dtype

_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512]


def _r32(v):
    return (v + 31) // 32 * 32


class ContentLoss(nn.Module):
    def __init__(self, feature_model_extractor_nodes: list, feature_model_normalize_mean: list,
                 feature_model_normalize_std: list, precision: Optional[str] = None,
                 inplace_relu_aliasing: bool = True) -> None:
        super().__init__()
        self.feature_model_extractor_nodes = feature_model_extractor_nodes
        self.precision = precision or os.environ.get("RESR_PRECISION", "fast")
        self._dtype = _precision_to_dtype(self.precision)
        self.aliasing = inplace_relu_aliasing
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
        """Pack all 16 convs once (frozen weights): per conv, cout groups of 64, K chunks of 32."""
        if self._packed is not None and self._packed[0].device == device:
            return
        L = _lib
        flat = torch.cat([getattr(self.features, str(l[1])).weight.detach().float().reshape(-1) for l in self.layers if l[0] == "conv"]).to(device)
        chunks, groups, off, src = [], {}, 0, 0
        for l in self.layers:
            if l[0] != "conv":
                continue
            _, idx, cin, cout = l
            gl = []
            for g0 in range(0, _r32(cout), 64):
                mt = min(64, _r32(cout) - g0) // 32
                gl.append((off, mt))
                for ck in range(_r32(cin) // 32):
                    chunks.append(L.PackChunk(src, off, cout, cin, g0, max(0, min(64, cout - g0)), ck * 32,
                                              max(0, min(32, cin - ck * 32)), mt, 0, 1.0, 0, None))
                    off += 9 * mt * 1024
            groups[idx] = gl
            src += cout * cin * 9
        host = (L.PackChunk * len(chunks))(*chunks)
        table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(device)
        es = 2 if self._dtype == L.RESR_F16 else 4
        packed = torch.zeros(off * es + 16384, dtype=torch.uint8, device=device)
        L.check(L.lib().resr_pack_weights(L.ptr(table), len(chunks), L.ptr(flat), L.ptr(packed), self._dtype, L.stream_ptr(flat)),
                "resr_pack_weights")
        self._packed = (packed, groups)

    def load_state_dict(self, state_dict, strict: bool = True):
        self._packed = None
        sd = {("features." + k[len("features."):]) if k.startswith("features.") else k: v for k, v in state_dict.items()}
        sd = {k: v for k, v in sd.items() if k.startswith("features.") and int(k.split(".")[1]) <= 34}
        sd.setdefault("mean", self.mean)
        sd.setdefault("std", self.std)
        return super().load_state_dict(sd, strict=strict)

    # ---- forward ---------------------------------------------------------------------------------------------
    def _features(self, x: torch.Tensor) -> dict:
        L, lib = _lib, _lib.lib()
        _lib.require_cuda(x, "ContentLoss")
        self._pack(x.device)
        packed, groups = self._packed
        st = L.stream_ptr(x)
        T = torch.float16 if self._dtype == L.RESR_F16 else torch.float32
        es = 2 if self._dtype == L.RESR_F16 else 4
        xn = ((x.float() - self.mean) / self.std).contiguous()              # transforms.Normalize, model.py:317-318
        n, _, h, w = xn.shape
        cur = torch.empty((n, h, w, 32), dtype=T, device=x.device)
        L.check(lib.resr_nchw_to_nhwc(L.ptr(xn), L.ptr(cur), n, 3, h, w, 1, 32, self._dtype, None, st))
        out = {}
        wanted = {int(k.split(".")[1]) for k in self.feature_model_extractor_nodes}
        last = max(wanted)
        for l in self.layers:
            if l[0] == "pool":
                c = cur.shape[-1]
                h, w = h // 2, w // 2
                nxt = torch.empty((n, h, w, c), dtype=T, device=x.device)
                L.check(lib.resr_maxpool2x2(L.ptr(cur), L.ptr(nxt), n, h, w, c, self._dtype, st))
                cur = nxt
                continue
            _, idx, cin, cout = l
            bias = getattr(self.features, str(idx)).bias
            tap_pre = idx in wanted and (not self.aliasing or idx == last)   # the value the extractor hands back
            relu_out = torch.empty((n, h, w, cout), dtype=T, device=x.device)
            pre = torch.empty_like(relu_out) if tap_pre else None
            flags = L.CONV_LRELU | (L.CONV_AUX_BEFORE_MASK if tap_pre else 0)   # slope 0 => ReLU; aux = pre-activation
            for gi, (off, mt) in enumerate(groups[idx]):
                g0 = gi * 64
                d = L.ConvDesc(n, h, w, _r32(cin), _r32(cin), cur.shape[-1], 0, min(mt * 32, cout - g0), mt * 32, cout,
                               0, 0, 0, self._dtype, flags, 1.0, 1.0, 1.0, 1.0, 0.0)
                L.check(lib.resr_conv3x3(C.byref(d), L.ptr(cur), None, C.c_void_p(packed.data_ptr() + off * es),
                                         C.c_void_p(bias.data_ptr() + g0 * 4), None, None, None,
                                         C.c_void_p(relu_out.data_ptr() + g0 * es),
                                         None if pre is None else C.c_void_p(pre.data_ptr() + g0 * es), st), "resr_conv3x3")
            if idx in wanted:
                out[f"features.{idx}"] = pre if tap_pre else relu_out
            cur = relu_out
            if idx == last:
                break
        return out

    @torch.no_grad()
    def forward(self, sr_tensor: torch.Tensor, hr_tensor: torch.Tensor):
        b = sr_tensor.shape[0]
        feats = self._features(torch.cat([sr_tensor.detach(), hr_tensor.detach()], 0))
        return tuple(F.l1_loss(feats[k][:b].float(), feats[k][b:].float()) for k in self.feature_model_extractor_nodes)
