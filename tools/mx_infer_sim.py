"""CPU emulation of the exact16 INFERENCE plans with MX-fp8 correction products (round 6).

Today's inference plan (x2_plan bits 0 + 5, "INFER40"): residual stream / HR tail chunks are hi + lo pairs against split weights --
three f16 stages x_hi W0 + x_hi W1 + x_lo W2 --, the growth planes o1..o4 single f16 tensors against f16 weights (one stage).  The MX
plan keeps the main product x_hi W0 on the f16 pipe and moves the two 2^-12-weighted corrections of every PAIR chunk into ONE
v_mfma_scale_f32_32x32x64_f8f6f4 per tap: A = [W1_8 | W2_8], B = [x_hi8 | x_lo8], K = 64 = the chunk's 32 channels twice.

What is emulated (forward only, float64 accumulation; the operand roundings are the only error source, as on the MFMA path):
  mx         e4m3 activations with one e8m0 scale per (pixel, 32-channel chunk, hi / lo), e4m3 weights with one scale per
             (cout, tap, 32-channel chunk, W1 / W2)
  mx_shared  as mx, but x_hi8 and x_lo8 of a pixel share ONE scale byte (block maximum of hi in [64, 128): x_lo * 2^12 <= 2 |x_hi|)
  bf8a       unscaled bf8 (e5m2) activations -- no scale tensor at all -- against MX e4m3 weights
  bf8        unscaled bf8 on both operands

    python tools/mx_infer_sim.py [--size 24] [--batch 1] [--seeds 11,12,13] [--wscale 1] [--json out.json]
Test infrastructure only (imports oracle/).
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_ref as M  # noqa: E402


def q16(t):
    return t.to(torch.float16).to(t.dtype)


def q8_mx(t, dim, top=7.0):
    """e4m3 with one power-of-two scale per 32 elements along `dim`; the block maximum lands in [2^top, 2^(top+1))."""
    t = t.movedim(dim, -1)
    shp = t.shape
    pad = (-shp[-1]) % 32
    tp = F.pad(t, (0, pad)) if pad else t
    b = tp.reshape(*tp.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -120)
    sc = torch.exp2(torch.floor(torch.log2(amax)) - top)
    q = (b / sc).to(torch.float32).to(torch.float8_e4m3fn).to(t.dtype) * sc
    return q.reshape(tp.shape)[..., :shp[-1]].movedim(-1, dim)


def q8_mx_by(t, ref, dim, top):
    """e4m3 of `t` with the block scales of `ref` (same shape): the shared scale byte of (x_hi8, x_lo8)."""
    t, ref = t.movedim(dim, -1), ref.movedim(dim, -1)
    shp = t.shape
    pad = (-shp[-1]) % 32
    if pad:
        t, ref = F.pad(t, (0, pad)), F.pad(ref, (0, pad))
    b, rb = t.reshape(*t.shape[:-1], -1, 32), ref.reshape(*t.shape[:-1], -1, 32)
    amax = rb.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -120)
    sc = torch.exp2(torch.floor(torch.log2(amax)) - top)
    q = (b / sc).to(torch.float32).clamp(-448, 448).to(torch.float8_e4m3fn).to(t.dtype) * sc
    return q.reshape(t.shape)[..., :shp[-1]].movedim(-1, dim)


def q8_e5m2(t, dim=None):
    return t.to(torch.float32).to(torch.float8_e5m2).to(t.dtype)


def conv_plan(x, w, b, pair_ch, mode):
    """One convolution of the inference plan: input channels [0, pair_ch) are pairs, the rest single f16 tensors against f16 weights."""
    xs, ws = x[:, :pair_ch], w[:, :pair_ch]
    if mode == "exact":
        y = F.conv2d(xs, ws, b, padding=1)
    else:
        xh = q16(xs)
        xl = xs - xh
        wh = q16(ws)
        wl = ws - wh
        y = F.conv2d(xh, wh, b, padding=1)
        if mode == "pairs":      # today: three f16 stages (x_lo and W1 are f16 values of the remainders)
            y = y + F.conv2d(xh, q16(wl * 4096.0) / 4096.0, None, padding=1) + F.conv2d(q16(xl * 4096.0) / 4096.0, wh, None, padding=1)
        elif mode == "mx":
            y = y + F.conv2d(q8_mx(xh, 1), q8_mx(wl, 1), None, padding=1) + F.conv2d(q8_mx(xl, 1), q8_mx(wh, 1), None, padding=1)
        elif mode == "mx_shared":
            y = y + F.conv2d(q8_mx_by(xh, xh, 1, 6.0), q8_mx(wl, 1), None, padding=1) + \
                F.conv2d(q8_mx_by(xl * 4096.0, xh, 1, 6.0) / 4096.0, q8_mx(wh, 1), None, padding=1)
        elif mode == "bf8a":
            y = y + F.conv2d(q8_e5m2(xh), q8_mx(wl, 1), None, padding=1) + F.conv2d(q8_e5m2(xl * 4096.0) / 4096.0, q8_mx(wh, 1), None, padding=1)
        elif mode == "bf8a_row":   # weights: e4m3 with ONE scale per (cout, 32-channel chunk, W1 / W2) shared by the nine taps
            def qrow(t):
                o, c = t.shape[:2]
                tt = t.reshape(o, c // 32 if c % 32 == 0 else 1, -1)
                amax = tt.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -120)
                sc = torch.exp2(torch.floor(torch.log2(amax)) - 7.0)
                return ((tt / sc).to(torch.float32).to(torch.float8_e4m3fn).to(t.dtype) * sc).reshape(t.shape)
            def qrow32(t):      # chunks of 32 input channels
                o, c = t.shape[:2]
                pad = (-c) % 32
                tp = F.pad(t, (0, 0, 0, 0, 0, pad)) if pad else t
                q = torch.cat([qrow(tp[:, k:k + 32]) for k in range(0, c + pad, 32)], 1)
                return q[:, :c]
            y = y + F.conv2d(q8_e5m2(xh), qrow32(wl), None, padding=1) + F.conv2d(q8_e5m2(xl * 4096.0) / 4096.0, qrow32(wh), None, padding=1)
        elif mode == "bf8":
            y = y + F.conv2d(q8_e5m2(xh), q8_e5m2(wl * 4096.0) / 4096.0, None, padding=1) + F.conv2d(q8_e5m2(xl * 4096.0) / 4096.0, q8_e5m2(wh), None, padding=1)
        elif mode == "hi":      # fast mode's product (for scale)
            pass
        else:
            raise ValueError(mode)
    if pair_ch < x.shape[1]:
        xg, wg = x[:, pair_ch:], w[:, pair_ch:]
        y = y + (F.conv2d(xg, wg, None, padding=1) if mode == "exact" else F.conv2d(q16(xg), q16(wg), None, padding=1))
    return y


def generator(x, sd, mode, upscale=4, n_blocks=23):
    """oracle.model_ref.generator_forward (model.py:255-272) under the inference plan."""
    lre = lambda t: F.leaky_relu(t, 0.2)   # noqa: E731
    c = lambda t, key, pair=None: conv_plan(t, sd[key + ".weight"], sd[key + ".bias"], t.shape[1] if pair is None else pair, mode)   # noqa: E731
    out1 = c(x, "conv1")
    t = out1
    for i in range(n_blocks):
        t0 = t
        for r in (1, 2, 3):
            feats = [t]
            for k in range(1, 5):
                o = lre(c(torch.cat(feats, 1), f"trunk.{i}.rdb{r}.conv{k}", 64))
                feats.append(o if mode == "exact" else q16(o))       # growth planes: single f16 tensors
            y5 = c(torch.cat(feats, 1), f"trunk.{i}.rdb{r}.conv5", 64)
            t = y5 * 0.2 + t if r < 3 else (y5 * 0.2 + t) * 0.2 + t0
    t = out1 + c(t, "conv2")
    t = lre(c(F.interpolate(t, scale_factor=2, mode="nearest"), "upsampling1.0"))
    t = lre(c(F.interpolate(t, scale_factor=2, mode="nearest"), "upsampling2.0"))
    t = lre(c(t, "conv3.0"))
    return torch.clamp(c(t, "conv4"), 0.0, 1.0)


def run(seed, size, batch, wscale, n_blocks=23, modes=("pairs", "mx", "bf8a", "bf8a_row", "bf8", "hi"), conv1_scale=1.0):
    sd = M.init_generator_state(seed, 3, 3, 4, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < n_blocks}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    if wscale != 1.0:
        sd = {k: (v * wscale if k.endswith(".weight") and ".rdb" in k else v) for k, v in sd.items()}
    if conv1_scale != 1.0:
        sd["conv1.weight"] = sd["conv1.weight"] * conv1_scale
    sd = {k: v.double() for k, v in sd.items()}
    x = torch.rand(batch, 3, size, size, generator=torch.Generator().manual_seed(5)).double()
    with torch.no_grad():
        y0 = generator(x, sd, "exact", 4, n_blocks)
        return {m: float((generator(x, sd, m, 4, n_blocks) - y0).abs().max()) for m in modes}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=23)
    ap.add_argument("--seeds", type=str, default="11,12,13")
    ap.add_argument("--wscale", type=str, default="1,4")
    ap.add_argument("--conv1-scale", type=float, default=1.0)
    ap.add_argument("--json", type=str, default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    out = {}
    for ws in [float(v) for v in a.wscale.split(",")]:
        for seed in [int(s) for s in a.seeds.split(",")]:
            r = run(seed, a.size, a.batch, ws, a.blocks, conv1_scale=a.conv1_scale)
            out[f"wscale{ws:g}_seed{seed}"] = r
            print(f"{a.batch} x {a.size}^2, dense weights x {ws:g}, conv1 x {a.conv1_scale:g}, seed {seed}: " + "  ".join(f"{k} {v:.2e}" for k, v in r.items()), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"geometry": f"{a.batch} x {a.size}^2, {a.blocks} blocks", "forward max-abs vs float64": out}, f, indent=1)
