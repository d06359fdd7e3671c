"""Is a step host-bound?  Time the ENQUEUE of N steps (no synchronisation) against the time until the device has finished them.
If enqueueing takes as long as the whole run, the host is the limiter.  Also times the host-only pieces of a step.

    python tools/host_bound.py [--gan | --cfg3] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import real_esrgan_pytorch_amd as R  # noqa: E402
from real_esrgan_pytorch_amd.degrade import Degrader, sample_plan  # noqa: E402
from real_esrgan_pytorch_amd.train import RealESRGANStep, RealESRNetStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gan", action="store_true")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--no-degradation", action="store_true")
a = ap.parse_args()
torch.manual_seed(0)
args = argparse.Namespace(noise_data=False)
if a.gan:
    B, tile, crop = 16, 400, 256
    g = R.Generator(3, 3, 4, precision="fast").cuda().train()
    d = R.Discriminator(precision="fast").cuda().train()
    ema = R.EMA(g, 0.999); ema.register()
    g_opt = torch.optim.Adam([g.flat_parameter()], 1e-4, (0.9, 0.99), fused=True)
    d_opt = torch.optim.Adam(d.parameters(), 1e-4, (0.9, 0.99), fused=True)
    content = R.ContentLoss(["features.2", "features.7", "features.16", "features.25", "features.34"], [0.485, 0.456, 0.406],
                            [0.229, 0.224, 0.225], precision="fast").cuda()
    hr = bench.make_hr_tiles(args, B, tile, 0)
    deg = None if a.no_degradation else Degrader(batch=B, hr_size=tile, upscale=4, crop=crop, seed=0)
    step_ = RealESRGANStep(g, d, ema, g_opt, d_opt, torch.amp.GradScaler("cuda"), deg, content_criterion=content)
    if deg is None:
        hrc = hr[:, :, :crop, :crop].contiguous(); lr = torch.nn.functional.interpolate(hrc, scale_factor=0.25, mode="area")
        step = lambda: step_(hrc, lr)
    else:
        step = lambda: step_(hr)
else:
    B, tile, crop = 32, 256, 256
    g = R.Generator(3, 3, 4, precision="fast").cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    ema = R.EMA(g, 0.999); ema.register()
    opt = torch.optim.Adam([g.flat_parameter()], 2e-4, (0.9, 0.99), fused=True)
    hr = bench.make_hr_tiles(args, B, tile, 0)
    deg = None if a.no_degradation else Degrader(batch=B, hr_size=tile, upscale=4, crop=crop, seed=0)
    step_ = RealESRNetStep(g, ema, opt, torch.amp.GradScaler("cuda"), deg)
    if deg is None:
        lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode="area")
        step = lambda: step_(hr, lr)
    else:
        step = lambda: step_(hr)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
t = time.perf_counter()
for _ in range(10):
    sample_plan(B, tile, tile, crop)
tp = (time.perf_counter() - t) / 10
print(f"{'gan' if a.gan else 'cfg3'} degradation={'off' if a.no_degradation else 'on'}: enqueue {1e3 * (t1 - t0) / a.steps:.2f} ms/step, until done {1e3 * (t2 - t0) / a.steps:.2f} ms/step; "
      f"sample_plan (host) {1e3 * tp:.2f} ms")
