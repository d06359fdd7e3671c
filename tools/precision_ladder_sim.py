"""CPU emulation of the precision rungs between `fast` (every operand one f16) and `exact16` (every operand an f16 pair).

The MFMA path multiplies f16 operands exactly and accumulates in fp32, so a rung is fully described by WHICH stored tensors
are single f16 and which are hi + lo pairs (~22 significand bits: emulated as "not rounded").  This script runs the oracle's
generator (oracle/model_ref.py restated with rounding hooks) in float64 with those roundings applied at the points where the
HIP path stores a tensor, forward and backward, and reports the two numbers the parity gate is about:

    forward max-abs vs the float64 evaluation, worst per-tensor relative L2 of the 702 weight gradients (+ input gradient)

so that only rungs that can pass are built as kernels.  Test infrastructure only (imports oracle/).

    python tools/precision_ladder_sim.py [--blocks 23] [--size 24] [--seeds 11,12,13] [--batch 1] [--wscale 1] [--set round5]
"""
import argparse
import math
import itertools
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_ref as M  # noqa: E402


def q16(t):
    return t.to(torch.float16).to(t.dtype)


def q8_mx(t, dim):
    """MX-style fp8: e4m3 values with one shared power-of-two scale per block of 32 elements along `dim` (the K dimension of
    the product the operand enters: input channels for conv / backward-data, pixels of a row for the weight gradients)."""
    t = t.movedim(dim, -1)
    shp = t.shape
    pad = (-shp[-1]) % 32
    tp = F.pad(t, (0, pad)) if pad else t
    b = tp.reshape(*tp.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    sc = torch.exp2(torch.floor(torch.log2(amax)) - 7.0)          # block maximum lands in [128, 256) of e4m3's 448
    q = (b / sc).to(torch.float32).to(torch.float8_e4m3fn).to(t.dtype) * sc
    q = q.reshape(tp.shape)[..., :shp[-1]]
    return q.movedim(-1, dim)


def q8_e5m2(t, dim=None):
    """bf8 (e5m2) WITHOUT block scales: f16's exponent range in one byte -- what an epilogue can emit next to the f16 value with one
    v_cvt_pk_bf8_f32 per two values, and what v_mfma_scale_f32_32x32x64_f8f6f4 multiplies with unit scales."""
    return t.to(torch.float32).to(torch.float8_e5m2).to(t.dtype)


def split16(t):
    hi = q16(t)
    return hi, t - hi


def q8_bf8s(t):
    """bf8 (e5m2, unit scales) of a REMAINDER operand as the HIP path stores it: times 2^12 (f16's range again), rounded, descaled."""
    return (t * 4096.0).to(torch.float32).to(torch.float8_e5m2).to(t.dtype) / 4096.0


def q8_row(t, kdim):
    """e4m3 WEIGHTS with ONE power-of-two scale per (output row of the product, 32-wide K chunk, all nine taps): what the packer emits
    for an MX stage.  `kdim` = the weight dimension the product contracts over (1: forward, 0: backward-data)."""
    tt = t.movedim(kdim, -1)                      # (rows, 3, 3, K)
    shp = tt.shape
    pad = (-shp[-1]) % 32
    tp = F.pad(tt, (0, pad)) if pad else tt
    b = tp.reshape(shp[0], 9, -1, 32).permute(0, 2, 1, 3)    # (rows, chunks, taps, 32)
    amax = b.abs().amax((-1, -2), keepdim=True).clamp_min(2.0 ** -120)
    sc = torch.exp2(torch.floor(torch.log2(amax)) - 7.0)
    q = (b / sc).to(torch.float32).to(torch.float8_e4m3fn).to(t.dtype) * sc
    q = q.permute(0, 2, 1, 3).reshape(tp.shape)[..., :shp[-1]]
    return q.movedim(-1, kdim)


class Conv8(torch.autograd.Function):
    """3x3 conv on split operands whose two 2^-12-weighted correction products take fp8 (MX e4m3) operands:
    x w ~ x_hi w_hi + q8(x_hi) q8(w_lo) + q8(x_lo) q8(w_hi), per pass selected by `where` ("f" forward, "d" backward-data,
    "w" weight gradients); passes not selected multiply the exact (pair) operands."""

    @staticmethod
    def forward(ctx, x, w, b, where):
        ctx.save_for_backward(x, w)
        ctx.where = where
        q8 = q8_e5m2 if "5" in where else q8_mx
        if "f" not in where:
            return F.conv2d(x, w, b, padding=1)
        xh, xl = split16(x)
        wh, wl = split16(w)
        if "6" in where:    # round 6: unscaled bf8 activations (lo times 2^12), row-scaled e4m3 weights
            return F.conv2d(xh, wh, b, padding=1) + F.conv2d(q8_e5m2(xh), q8_row(wl, 1), None, padding=1) + F.conv2d(q8_bf8s(xl), q8_row(wh, 1), None, padding=1)
        if "7" in where:    # round 6, as built: unscaled bf8 on BOTH operands (remainders times 2^12) -- no scale bytes anywhere
            return F.conv2d(xh, wh, b, padding=1) + F.conv2d(q8_e5m2(xh), q8_bf8s(wl), None, padding=1) + F.conv2d(q8_bf8s(xl), q8_e5m2(wh), None, padding=1)
        return F.conv2d(xh, wh, b, padding=1) + F.conv2d(q8(xh, 1), q8(wl, 1), None, padding=1) + F.conv2d(q8(xl, 1), q8(wh, 1), None, padding=1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        q8 = q8_e5m2 if "5" in ctx.where else q8_mx
        gh, gl = split16(g)
        wh, wl = split16(w)
        xh, xl = split16(x)
        ci = lambda gg, ww: torch.nn.grad.conv2d_input(x.shape, ww, gg, padding=1)    # noqa: E731
        cw = lambda xx, gg: torch.nn.grad.conv2d_weight(xx, w.shape, gg, padding=1)   # noqa: E731
        # backward-data contracts over the OUTPUT channels (dim 1 of g, dim 0 of w)
        if "7" in ctx.where:
            gx = ci(gh, wh) + ci(q8_e5m2(gh), q8_bf8s(wl)) + ci(q8_bf8s(gl), q8_e5m2(wh)) if "d" in ctx.where else ci(g, w)
            gw = cw(xh, gh) + cw(q8_e5m2(xh), q8_bf8s(gl)) + cw(q8_bf8s(xl), q8_e5m2(gh)) if "w" in ctx.where else cw(x, g)
            return gx, gw, g.sum((0, 2, 3)), None
        if "6" in ctx.where:
            gx = ci(gh, wh) + ci(q8_e5m2(gh), q8_row(wl, 0)) + ci(q8_bf8s(gl), q8_row(wh, 0)) if "d" in ctx.where else ci(g, w)
            gw = cw(xh, gh) + cw(q8_e5m2(xh), q8_bf8s(gl)) + cw(q8_bf8s(xl), q8_e5m2(gh)) if "w" in ctx.where else cw(x, g)
            return gx, gw, g.sum((0, 2, 3)), None
        gx = ci(gh, wh) + ci(q8(gh, 1), q8(wl, 0)) + ci(q8(gl, 1), q8(wh, 0)) if "d" in ctx.where else ci(g, w)
        # the weight gradients contract over pixels: blocks of 32 along a row
        gw = cw(xh, gh) + cw(q8(xh, 3), q8(gl, 3)) + cw(q8(xl, 3), q8(gh, 3)) if "w" in ctx.where else cw(x, g)
        return gx, gw, g.sum((0, 2, 3)), None


class Store(torch.autograd.Function):
    """A tensor the HIP path stores: value rounded to `fa` on the way forward, its gradient to `fg` on the way back."""

    @staticmethod
    def forward(ctx, x, fa, fg):
        ctx.fg = fg
        return q16(x) if fa == "f16" else x

    @staticmethod
    def backward(ctx, g):
        return (q16(g) if ctx.fg == "f16" else g), None, None


class Conv(torch.autograd.Function):
    """3x3 conv whose weight operand is f16 or split (exact), and whose weight-gradient operands (x, g) may be rounded to
    f16 separately from the data path (exact16's hi-only weight gradients)."""

    @staticmethod
    def forward(ctx, x, w, b, fw, fwg, fwb, gread="pair", wx_from=None, wxg_hi=False, w16_from=None):
        wq = q16(w) if fw == "f16" else w
        if w16_from is not None:     # forward only: the input chunks from w16_from on (the growth planes) meet f16 weights -- ONE stage, x W0
            wq = torch.cat([wq[:, :w16_from], q16(wq[:, w16_from:])], 1)
        ctx.save_for_backward(x, w)
        ctx.fwg, ctx.fwb, ctx.gread, ctx.wx_from, ctx.wxg_hi = fwg, fwb, gread, wx_from, wxg_hi
        return F.conv2d(x, wq, b, padding=1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        # gread = "hi": the stored gradient plane is a PAIR, but backward-data and the weight products read its hi tensor only;
        # the bias sum (and nothing else) takes hi + lo
        gr = q16(g) if ctx.gread == "hi" else g
        gx = torch.nn.grad.conv2d_input(x.shape, q16(w) if ctx.fwb == "f16" else w, gr, padding=1)
        fx, fg = ctx.fwg if isinstance(ctx.fwg, tuple) else (ctx.fwg, ctx.fwg)
        xw = q16(x) if fx == "f16" else x
        if ctx.wx_from is not None:      # the weight products read the input channels from wx_from on (the growth planes) as their hi tensor only
            xw = torch.cat([xw[:, :ctx.wx_from], q16(xw[:, ctx.wx_from:])], 1)
        gq = q16(gr) if fg == "f16" else gr
        if ctx.wx_from is not None and ctx.wxg_hi:   # ... and those chunks' products take G's hi tensor too: one tap-product, (x_hi, g_hi)
            k = ctx.wx_from
            gw = torch.cat([torch.nn.grad.conv2d_weight(xw[:, :k], (w.shape[0], k, 3, 3), gq, padding=1),
                            torch.nn.grad.conv2d_weight(xw[:, k:], (w.shape[0], w.shape[1] - k, 3, 3), q16(gq), padding=1)], 1)
        else:
            gw = torch.nn.grad.conv2d_weight(xw, w.shape, gq, padding=1)
        return gx, gw, g.sum((0, 2, 3)), None, None, None, None, None, None, None


def generator(x, sd, cfg, upscale=4, n_blocks=23):
    """oracle.model_ref.generator_forward (model.py:255-272) with the storage roundings of `cfg`."""
    S = lambda t, cls: Store.apply(t, cfg["a_" + cls], cfg["g_" + cls])   # noqa: E731

    def conv(t, key):
        if cfg.get("fp8"):
            return Conv8.apply(t, sd[key + ".weight"], sd[key + ".bias"], cfg["fp8"])
        growth = ".rdb" in key and not key.endswith("conv5")      # conv1..4 of a dense block: their G operand is a growth-plane gradient
        wg = cfg.get("wg_growth", cfg["wg"]) if growth else cfg["wg"]
        wx_from = 64 if (".rdb" in key and ((cfg.get("wx5_growth") == "hi" and key.endswith("conv5")) or
                                             (cfg.get("wx_growth") == "hi" and not key.endswith("conv1")))) else None
        wb = cfg.get("wb_growth", cfg.get("wb", cfg["w"])) if growth else cfg.get("wb", cfg["w"])   # backward-data weights of conv1..4: their G is the single chunk of the mirrored passes
        return Conv.apply(t, sd[key + ".weight"], sd[key + ".bias"], cfg["w"], wg, wb,
                          cfg.get("gread_growth", "pair") if growth else "pair", wx_from, bool(cfg.get("wxg5_hi")) and key.endswith("conv5"),
                          64 if (cfg.get("w_growth_fwd") == "f16" and ".rdb" in key and not key.endswith("conv1")) else None)

    x = S(x, "in")
    out1 = S(conv(x, "conv1"), "stream")
    t = out1
    for i in range(n_blocks):
        t0 = t
        for r in (1, 2, 3):
            feats = [t]
            for c in range(1, 5):
                feats.append(S(F.leaky_relu(conv(torch.cat(feats, 1), f"trunk.{i}.rdb{r}.conv{c}"), 0.2), "dense"))
            t = S(conv(torch.cat(feats, 1), f"trunk.{i}.rdb{r}.conv5") * 0.2 + t, "stream") if r < 3 else \
                S((conv(torch.cat(feats, 1), f"trunk.{i}.rdb{r}.conv5") * 0.2 + t) * 0.2 + t0, "stream")
    t = S(out1 + conv(t, "conv2"), "stream")
    t = S(F.leaky_relu(conv(F.interpolate(t, scale_factor=2, mode="nearest"), "upsampling1.0"), 0.2), "tail")
    t = S(F.leaky_relu(conv(F.interpolate(t, scale_factor=2, mode="nearest"), "upsampling2.0"), 0.2), "tail")
    t = S(F.leaky_relu(conv(t, "conv3.0"), 0.2), "tail")
    return torch.clamp(conv(t, "conv4"), 0.0, 1.0)


def mk(a_stream, a_dense, a_tail, w, wg, g_stream=None, g_dense=None, g_tail=None, a_in=None, wb=None):
    return {"wb": wb or w, "a_stream": a_stream, "a_dense": a_dense, "a_tail": a_tail, "a_in": a_in or a_stream,
            "g_stream": g_stream or a_stream, "g_dense": g_dense or a_dense, "g_tail": g_tail or a_tail, "g_in": "pair",
            "w": w, "wg": wg}


RUNGS = {
    "fast (all f16)":                                   mk("f16", "f16", "f16", "f16", "f16"),
    "W split only (x_hi W0 + x_hi W1)":                 mk("f16", "f16", "f16", "split", "f16"),
    "acts pair, W f16 (x_hi W + x_lo W)":               mk("pair", "pair", "pair", "f16", "f16"),
    "stream pair, dense+tail f16, W f16":               mk("pair", "f16", "f16", "f16", "f16"),
    "stream pair, dense+tail f16, W split":             mk("pair", "f16", "f16", "split", "f16"),
    "stream+tail pair, dense f16, W f16":               mk("pair", "f16", "pair", "f16", "f16"),
    "stream+tail pair, dense f16, W split":             mk("pair", "f16", "pair", "split", "f16"),
    "stream+tail pair, dense f16 fwd / pair bwd, W split": mk("pair", "f16", "pair", "split", "f16", g_dense="pair"),
    "fwd exact; bwd: g f16 all, W split (2 products)":   mk("pair", "pair", "pair", "split", ("pair", "f16"), g_stream="f16", g_dense="f16", g_tail="f16"),
    "fwd exact; bwd: g f16 all, W f16 (1 product)":      mk("pair", "pair", "pair", "split", ("pair", "f16"), g_stream="f16", g_dense="f16", g_tail="f16", wb="f16"),
    "fwd exact; bwd: g_dense f16, g_stream/tail pair, W split": mk("pair", "pair", "pair", "split", ("pair", "f16"), g_dense="f16"),
    "fwd exact; bwd: g pair, W f16 in bwd-data (2 products)":   mk("pair", "pair", "pair", "split", "pair", wb="f16"),
    "exact16, wgrad x_hi (g pair): 2 products":          mk("pair", "pair", "pair", "split", ("f16", "pair")),
    "exact16, wgrad g_hi (x pair): 2 products":          mk("pair", "pair", "pair", "split", ("pair", "f16")),
    "exact16 (all pair, W split, wgrad hi-only)":       mk("pair", "pair", "pair", "split", "f16"),
    "exact16x3 (all pair, W split, wgrad pairs)":       mk("pair", "pair", "pair", "split", "pair"),
}


def _with(cfg, **kw):
    c = dict(cfg)
    c.update(kw)
    return c


_EXACT = mk("pair", "pair", "pair", "split", "pair")
# round 5: the two rungs that get built, as they are built, + fp8 (MX e4m3) operands for the 2^-12-weighted correction products
RUNGS5 = {
    "INFER: stream+tail pair, growth planes f16, W split":           mk("pair", "f16", "pair", "split", "pair"),
    "INFER40: INFER + the growth chunks meet f16 weights (ONE stage each: 40 stages per block)": _with(mk("pair", "f16", "pair", "split", "pair"), w_growth_fwd="f16"),
    "TRAIN: fwd exact; growth-plane gradients f16 (bwd-data 2 stages, wgrad conv1-4 2 products)": mk("pair", "pair", "pair", "split", "pair", g_dense="f16"),
    "TRAIN2: growth-plane gradients stored as pairs, READ as hi only by backward-data and the weight products; bias sums from hi + lo": _with(_EXACT, gread_growth="hi"),
    "TRAIN3: TRAIN2 + conv5's weight products read the growth planes o1..o4 as their hi tensor": _with(_EXACT, gread_growth="hi", wx5_growth="hi"),
    "TRAIN4: TRAIN2 + EVERY weight product reads the growth planes o1..o4 as their hi tensor (conv2..conv5)": _with(_EXACT, gread_growth="hi", wx_growth="hi"),
    "TRAIN5: TRAIN4 + conv5's growth-plane products take g_y's hi tensor too (one tap-product per growth chunk)": _with(_EXACT, gread_growth="hi", wx_growth="hi", wxg5_hi=True),
    "TRAIN6: TRAIN5 + backward-data multiplies the growth-plane gradients with f16 weights (ONE stage per single chunk: 40 stages)": _with(_EXACT, gread_growth="hi", wx_growth="hi", wxg5_hi=True, wb_growth="f16"),
    "TRAIN + hi-only wgrad on conv1-4 only":                         _with(mk("pair", "pair", "pair", "split", "pair", g_dense="f16"), wg_growth="f16"),
    "fp8 corrections: weight gradients only":                        _with(_EXACT, fp8="w"),
    "fp8 corrections: backward-data + weight gradients":             _with(_EXACT, fp8="dw"),
    "fp8 corrections: forward + backward-data + weight gradients":   _with(_EXACT, fp8="fdw"),
    "fp8 corrections: forward only":                                 _with(_EXACT, fp8="f"),
    "bf8 (e5m2, unit scales) corrections: weight gradients only":    _with(_EXACT, fp8="w5"),
    "bf8 (e5m2, unit scales) corrections: backward-data only":       _with(_EXACT, fp8="d5"),
    "bf8 (e5m2, unit scales) corrections: backward-data + weight gradients": _with(_EXACT, fp8="dw5"),
    "bf8x4096 acts + row-scaled e4m3 weights (round 6): backward-data only":            _with(_EXACT, fp8="d6"),
    "bf8x4096 acts (round 6): weight gradients only":                                   _with(_EXACT, fp8="w6"),
    "bf8x4096 acts + row-scaled e4m3 weights (round 6): backward-data + weight gradients": _with(_EXACT, fp8="dw6"),
    "bf8x4096 on both operands (round 6, as built): backward-data only":                 _with(_EXACT, fp8="d7"),
    "bf8x4096 on both operands (round 6, as built): backward-data + weight gradients":   _with(_EXACT, fp8="dw7"),
    "exact16x3 (all pair, W split, wgrad pairs)":                    _EXACT,
}


def run(seed, n_blocks, size, upscale=4, only=None, batch=1, wscale=1.0, rungs=None, loss="dense"):
    sd = M.init_generator_state(seed, 3, 3, upscale, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < n_blocks}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    if wscale != 1.0:   # off the init scale: every convolution weight times `wscale` (the dense branches grow with it)
        sd = {k: (v * wscale if k.endswith(".weight") and ".rdb" in k else v) for k, v in sd.items()}
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(batch, 3, size, size, generator=gen).double()
    gw = torch.randn(batch, 3, size * upscale, size * upscale, generator=gen).double()
    exact = mk("pair", "pair", "pair", "split", "pair")
    # loss = "l1": the train step's own loss, mean |y - target| (its gradient has ONE magnitude, 1 / numel, and pulls coherently); the
    # rungs then run at the loss scale the product's power-of-two lift would put them at (max |g_y| in [2^6, 2^7): generator.hip)
    target = torch.rand(batch, 3, size * upscale, size * upscale, generator=gen).double()
    rung_scale = 1024.0 if loss == "dense" else 2.0 ** math.ceil(math.log2(64.0 * gw.numel()))

    def one(cfg, scale):
        p = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
        xi = x.clone().requires_grad_(True)
        y = generator(xi, p, cfg, upscale, n_blocks)
        ((y * gw).sum() if loss == "dense" else (y - target).abs().mean()).mul(scale).backward()
        return y.detach(), {k: v.grad / scale for k, v in p.items()}, xi.grad / scale
    y0, g0, gx0 = one(exact, 1.0)
    rows = {}
    for name, cfg in (rungs or RUNGS).items():
        if only and not any(o in name for o in only):
            continue
        y, g, gx = one(cfg, rung_scale)
        rel = {k: ((g[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30)).item() for k in g0}
        worst_k = max(rel, key=rel.get)
        vals = sorted(rel.values())
        wrel = {k: v for k, v in rel.items() if k.endswith(".weight")}
        worst_w = max(wrel, key=wrel.get)
        rows[name] = {"fwd_max_abs": (y - y0).abs().max().item(), "act_max": float(y0.abs().max()), "grad_worst_weight": wrel[worst_w],
                      "grad_worst_weight_tensor": worst_w, "grad_worst": rel[worst_k], "grad_worst_tensor": worst_k,
                      "grad_median": vals[len(vals) // 2], "gx": ((gx - gx0).norm() / gx0.norm()).item(),
                      "grad_worst_conv5": max(v for k, v in rel.items() if ".conv5." in k)}
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=23)
    ap.add_argument("--size", type=int, default=24)
    ap.add_argument("--seeds", type=str, default="11")
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--json", type=str, default="")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--wscale", type=float, default=1.0, help="dense-block weights times this (off the init scale)")
    ap.add_argument("--set", type=str, default="ladder", choices=["ladder", "round5"])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--loss", type=str, default="dense", choices=["dense", "l1"], help="dense: a random cotangent (the worst case); l1: the train step's mean |y - target|")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    allrows = {}
    for seed in [int(s) for s in a.seeds.split(",")]:
        rows = run(seed, a.blocks, a.size, only=[o for o in a.only.split(",") if o], batch=a.batch, wscale=a.wscale,
                   rungs=RUNGS5 if a.set == "round5" else RUNGS, loss=a.loss)
        allrows[seed] = rows
        print(f"== seed {seed}, {a.blocks} blocks, {a.batch} x {a.size}^2 LR, dense weights x {a.wscale}", flush=True)
        for name, r in rows.items():
            print(f"{name:58s} fwd {r['fwd_max_abs']:.2e}  grad worst {r['grad_worst']:.2e} ({r['grad_worst_tensor']})  worst weight {r['grad_worst_weight']:.2e}  worst conv5 {r['grad_worst_conv5']:.2e}  median {r['grad_median']:.2e}  gx {r['gx']:.2e}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(allrows, f, indent=1)
