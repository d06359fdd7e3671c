"""Oracle restatement of the reference networks (reference: model.py).

Functional, state_dict-driven, CPU fp32.  Every function cites the reference lines it
follows.  Key names are the reference's.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

SLOPE = 0.2  # LeakyReLU negative slope, model.py:81


# --------------------------------------------------------------------------------------
# initialisation (model.py:100-106 for dense blocks; torch.nn.Conv2d default elsewhere)
# --------------------------------------------------------------------------------------
def _default_conv_init(cout: int, cin: int, k: int, gen: torch.Generator, bias: bool = True):
    """torch.nn.Conv2d.reset_parameters: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +)."""
    fan_in = cin * k * k
    bound = 1.0 / math.sqrt(fan_in)
    w = (torch.rand(cout, cin, k, k, generator=gen) * 2 - 1) * bound
    b = (torch.rand(cout, generator=gen) * 2 - 1) * bound if bias else None
    return w, b


def generator_conv_shapes(in_channels: int = 3, out_channels: int = 3, upscale_factor: int = 4):
    """Ordered (key, cout, cin) list of the generator's 351 convs (model.py:207-252)."""
    cin0 = in_channels * {4: 1, 2: 4, 1: 16}[upscale_factor]
    shapes = [("conv1", 64, cin0)]
    for i in range(23):
        for r in (1, 2, 3):
            for c in range(1, 6):
                shapes.append((f"trunk.{i}.rdb{r}.conv{c}", 32 if c < 5 else 64, 64 + 32 * (c - 1)))
    shapes += [("conv2", 64, 64), ("upsampling1.0", 64, 64), ("upsampling2.0", 64, 64),
               ("conv3.0", 64, 64), ("conv4", out_channels, 64)]
    return shapes


def init_generator_state(seed: int = 0, in_channels: int = 3, out_channels: int = 3,
                         upscale_factor: int = 4, rdb_scale: float = 0.1,
                         bias_noise: float = 0.0) -> Dict[str, torch.Tensor]:
    """Random-init generator weights with the reference's *distributions*
    (dense-block convs: kaiming_normal * 0.1, bias 0 -- model.py:100-106; the rest: Conv2d default).
    `bias_noise` > 0 perturbs the zero biases so parity tests exercise the bias path."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for key, cout, cin in generator_conv_shapes(in_channels, out_channels, upscale_factor):
        if ".rdb" in key:
            std = math.sqrt(2.0 / (cin * 9))
            w = torch.randn(cout, cin, 3, 3, generator=gen) * std * rdb_scale
            b = torch.zeros(cout)
        else:
            w, b = _default_conv_init(cout, cin, 3, gen)
        if bias_noise > 0:
            b = b + torch.randn(cout, generator=gen) * bias_noise
        sd[key + ".weight"] = w
        sd[key + ".bias"] = b
    return sd


# --------------------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------------------
def _conv(x, sd, key, padding=1, stride=1):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding)


def rdb_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """ResidualDenseBlock.forward, model.py:87-98."""
    feats = [x]
    for c in range(1, 5):
        feats.append(F.leaky_relu(_conv(torch.cat(feats, 1), sd, f"{prefix}.conv{c}"), SLOPE))
    o5 = _conv(torch.cat(feats, 1), sd, f"{prefix}.conv5")
    return o5 * 0.2 + x


def rrdb_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """ResidualResidualDenseBlock.forward, model.py:123-132."""
    y = x
    for r in (1, 2, 3):
        y = rdb_forward(y, sd, f"{prefix}.rdb{r}")
    return y * 0.2 + x


def generator_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], upscale_factor: int = 4,
                      n_blocks: int = 23) -> torch.Tensor:
    """Generator._forward_impl, model.py:255-272 (pixel-unshuffle for x2/x1: model.py:209-220)."""
    down = {4: 1, 2: 2, 1: 4}[upscale_factor]
    y = F.pixel_unshuffle(x, down) if down > 1 else x
    out1 = _conv(y, sd, "conv1")
    t = out1
    for i in range(n_blocks):
        t = rrdb_forward(t, sd, f"trunk.{i}")
    t = out1 + _conv(t, sd, "conv2")
    t = F.leaky_relu(_conv(F.interpolate(t, scale_factor=2, mode="nearest"), sd, "upsampling1.0"), SLOPE)
    t = F.leaky_relu(_conv(F.interpolate(t, scale_factor=2, mode="nearest"), sd, "upsampling2.0"), SLOPE)
    t = F.leaky_relu(_conv(t, sd, "conv3.0"), SLOPE)
    t = _conv(t, sd, "conv4")
    return torch.clamp(t, 0.0, 1.0)


# --------------------------------------------------------------------------------------
# discriminator (model.py:135-203) with torch.nn.utils.spectral_norm semantics
# --------------------------------------------------------------------------------------
DISC_SN_LAYERS = (("down_block1", 128, 64, 4), ("down_block2", 256, 128, 4), ("down_block3", 512, 256, 4),
                  ("up_block1", 256, 512, 3), ("up_block2", 128, 256, 3), ("up_block3", 64, 128, 3),
                  ("conv2", 64, 64, 3), ("conv3", 64, 64, 3))


def _l2n(v: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    return v / v.norm().clamp_min(eps)


def init_discriminator_state(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Conv2d default init + spectral_norm's u/v init (normal, normalised); key names per SURVEY §8b."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    w, b = _default_conv_init(64, 3, 3, gen)
    sd["conv1.weight"], sd["conv1.bias"] = w, b
    for name, cout, cin, k in DISC_SN_LAYERS:
        w, _ = _default_conv_init(cout, cin, k, gen, bias=False)
        sd[f"{name}.0.weight_orig"] = w
        sd[f"{name}.0.weight_u"] = _l2n(torch.randn(cout, generator=gen))
        sd[f"{name}.0.weight_v"] = _l2n(torch.randn(cin * k * k, generator=gen))
    w, b = _default_conv_init(1, 64, 3, gen)
    sd["conv4.weight"], sd["conv4.bias"] = w, b
    return sd


def spectral_norm_weight(sd: Dict[str, torch.Tensor], name: str, training: bool,
                         eps: float = 1e-12) -> torch.Tensor:
    """torch.nn.utils.spectral_norm (hook form, used at model.py:140-168): one power iteration
    per training-mode forward, u/v updated in place without grad; sigma = u . (W v) with grad
    through W only; W = W_orig / sigma."""
    w = sd[f"{name}.0.weight_orig"]
    u, v = sd[f"{name}.0.weight_u"], sd[f"{name}.0.weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v_new = _l2n(torch.mv(wm.t(), u), eps)
            u_new = _l2n(torch.mv(wm, v_new), eps)
            u.copy_(u_new)
            v.copy_(v_new)
    sigma = torch.dot(u.detach().clone(), torch.mv(wm, v.detach().clone()))
    return w / sigma


def discriminator_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], training: bool = True) -> torch.Tensor:
    """Discriminator._forward_impl, model.py:177-203."""
    def sn(name, inp, stride, pad):
        w = spectral_norm_weight(sd, name, training)
        return F.leaky_relu(F.conv2d(inp, w, None, stride=stride, padding=pad), SLOPE)

    def up(t):
        return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)

    out1 = _conv(x, sd, "conv1")
    d1 = sn("down_block1", out1, 2, 1)
    d2 = sn("down_block2", d1, 2, 1)
    d3 = sn("down_block3", d2, 2, 1)
    u1 = sn("up_block1", up(d3), 1, 1) + d2
    u2 = sn("up_block2", up(u1), 1, 1) + d1
    u3 = sn("up_block3", up(u2), 1, 1) + out1
    o = sn("conv2", u3, 1, 1)
    o = sn("conv3", o, 1, 1)
    return _conv(o, sd, "conv4")


# --------------------------------------------------------------------------------------
# EMA (model.py:30-61)
# --------------------------------------------------------------------------------------
def ema_update(shadow: Dict[str, torch.Tensor], params: Dict[str, torch.Tensor], decay: float) -> None:
    """EMA.update, model.py:43-48: shadow = (1-decay)*p + decay*shadow (this operand order)."""
    for k, p in params.items():
        shadow[k] = (1.0 - decay) * p + decay * shadow[k]


# --------------------------------------------------------------------------------------
# ContentLoss (model.py:278-335): torchvision VGG19 `features` graph restated; weights are supplied
# (pretrained weights are unavailable offline -> numerics vs torchvision itself are unpinned)
# --------------------------------------------------------------------------------------
VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512]


def vgg_features(x, sd, nodes, mean, std, inplace_relu_aliasing: bool = True):
    """The tapped nodes of `vgg19().features` the way `create_feature_extractor` hands them back (model.py:296-303,:317-323).
    `inplace_relu_aliasing`: torchvision's ReLU(inplace=True) overwrites every tapped conv output except the last one (the
    extractor prunes the graph after it)."""
    m = torch.tensor(mean, dtype=x.dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=x.dtype).view(1, 3, 1, 1)
    wanted = {int(k.split(".")[1]) for k in nodes}
    last = max(wanted)
    x = (x - m) / s
    out, idx = {}, 0
    for v in VGG19_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2)
            idx += 1
            continue
        pre = F.conv2d(x, sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"], padding=1)
        x = F.relu(pre)
        if idx in wanted:
            out[f"features.{idx}"] = pre if (not inplace_relu_aliasing or idx == last) else x
        if idx == last:
            break
        idx += 2
    return out


def content_loss(sr, hr, sd, nodes, mean, std, inplace_relu_aliasing: bool = True):
    """Five L1 feature losses (model.py:311-335)."""
    a = vgg_features(sr, sd, nodes, mean, std, inplace_relu_aliasing)
    b = vgg_features(hr, sd, nodes, mean, std, inplace_relu_aliasing)
    return tuple(F.l1_loss(a[k], b[k]) for k in nodes)
