"""Upper bound of "bilinear x2 inside the conv producer" (the discriminator's three up blocks are
bilinear x2 -> 3x3 conv, model.py:190-199; today the interpolation is its own launch, csrc/disc.hip bilinear_up_quad_kernel).

A fused gather would let the conv's producer waves read the LOW-resolution tensor and interpolate on the way into LDS: the
up-sampled tensor (4x the pixels) would be neither written nor read.  The cheapest honest experiment is a bound: the same conv
launched with RESR_CONV_UPSAMPLE_IN on the low-resolution tensor -- the producers' NEAREST x2 gather (wrong values for this
layer, right memory side: a quarter of the input bytes, no intermediate tensor) -- against the two launches of today.  What a
real fusion would still owe on top of the bound: four source pixels per LDS piece instead of one LDS-DMA copy, i.e. vector
loads + 3 packed lerps + ds_write on the producer waves, which share their SIMDs' issue ports with the MFMA waves.

    python tools/bilinear_fuse_bound.py [--json profiles/r05_bilinear_fuse_bound.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import real_esrgan_pytorch_amd as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402

L = R._lib


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    lib = L.lib()
    n = 16
    rows = []
    # the three up blocks of the U-Net at 16 x 256^2 (config 4): (cin, cout, low-res edge)
    for cin, cout, e in ((512, 256, 32), (256, 128, 64), (128, 64, 128)):
        g = torch.Generator().manual_seed(cin)
        lo = (torch.randn(n, e, e, cin, generator=g) * 0.5).half().cuda()
        hi = torch.empty(n, 2 * e, 2 * e, cin, dtype=torch.float16, device="cuda")
        out = torch.empty(n, 2 * e, 2 * e, cout, dtype=torch.float16, device="cuda")
        wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        groups = cout // 64
        per = (cin // 32) * 9 * 2 * 1024 * 2
        packed = torch.cat([U.pack_conv(wt[q * 64:(q + 1) * 64], L.RESR_F16)[:per] for q in range(groups)] + [torch.zeros(16384, dtype=torch.uint8, device="cuda")])
        st = L.stream_ptr()

        def desc(flags):
            d = L.ConvDesc(n, 2 * e, 2 * e, cin, cin, cin, 0, 64, 64, cout, 0, 0, 0, L.RESR_F16, flags | L.CONV_LRELU | L.CONV_NO_BIAS, 1.0, 1.0, 1.0, 1.0, 0.2)
            d.cout_groups = groups
            return d
        d_hi, d_lo = desc(0), desc(L.CONV_UPSAMPLE_IN)

        def bil():
            L.check(lib.resr_bilinear_up2x(L.ptr(lo), L.ptr(hi), n, e, e, cin, L.RESR_F16, 0, st), "bilinear")

        def conv_hi():
            L.check(lib.resr_conv3x3(C.byref(d_hi), L.ptr(hi), None, L.ptr(packed), None, None, None, None, L.ptr(out), None, st), "conv")

        def conv_lo():
            L.check(lib.resr_conv3x3(C.byref(d_lo), L.ptr(lo), None, L.ptr(packed), None, None, None, None, L.ptr(out), None, st), "conv")
        t_b, t_c, t_f = timed(bil), timed(conv_hi), timed(conv_lo)
        t_both = timed(lambda: (bil(), conv_hi()))
        row = {"layer": f"{cin}->{cout} @ {2 * e}^2 x {n}", "bilinear_us": round(t_b, 1), "conv_us": round(t_c, 1), "both_back_to_back_us": round(t_both, 1),
               "conv_with_gather_on_low_res_us": round(t_f, 1), "bound_saving_us": round(t_both - t_f, 1),
               "intermediate_mb": round(hi.numel() * 2 / 1e6, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    tot = sum(r["bound_saving_us"] for r in rows)
    print(f"upper bound per discriminator forward: {tot:.1f} us; three forwards per RealESRGAN step: {3 * tot / 1e3:.2f} ms")
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"rows": rows, "bound_per_forward_us": tot}, f, indent=1)


if __name__ == "__main__":
    main()
