"""Which torch ops launch the small kernels of a RealESRGAN step?  torch.profiler over a few steps of bench.py's GAN step:
op counts per step (CPU side) and kernel counts per step (device side).

    python tools/prof_gan_ops.py [--steps 3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.degrade import Degrader
    from real_esrgan_pytorch_amd.train import RealESRGANStep
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="fast").cuda().train()
    d = R.Discriminator(precision="fast").cuda().train()
    ema = R.EMA(g, 0.999)
    ema.register()
    g_opt = torch.optim.Adam([g.flat_parameter()], 1e-4, (0.9, 0.99), fused=True)
    d_opt = torch.optim.Adam(d.parameters(), 1e-4, (0.9, 0.99), fused=True)
    content = R.ContentLoss(["features.2", "features.7", "features.16", "features.25", "features.34"], [0.485, 0.456, 0.406],
                            [0.229, 0.224, 0.225], precision="fast").cuda()
    args = argparse.Namespace(noise_data=False)
    hr = bench.make_hr_tiles(args, 16, 400, 0)
    degrade = Degrader(batch=16, hr_size=400, upscale=4, crop=256, seed=0)
    step = RealESRGANStep(g, d, ema, g_opt, d_opt, torch.amp.GradScaler("cuda"), degrade, content_criterion=content)
    for _ in range(5):
        step(hr)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(a.steps):
            step(hr)
        torch.cuda.synchronize()
    ev = prof.key_averages()
    rows = sorted(ev, key=lambda e: -e.count)
    print(f"{'op / kernel':80s} {'per step':>9s} {'cpu us/call':>12s} {'dev us/call':>12s}")
    for e in rows[:70]:
        dev = getattr(e, "device_time", getattr(e, "cuda_time", 0.0))
        print(f"{e.key[:80]:80s} {e.count / a.steps:9.1f} {e.cpu_time:12.1f} {dev:12.1f}")
    print()
    print(f"{'kernels by device time':110s} {'per step':>9s} {'us/call':>9s} {'ms/step':>9s}")
    def dev_total(e):
        return getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0.0))
    krows = [e for e in ev if dev_total(e) > 0 and e.cpu_time_total == 0]
    for e in sorted(krows, key=lambda e: -dev_total(e))[:45]:
        print(f"{e.key[:110]:110s} {e.count / a.steps:9.1f} {dev_total(e) / e.count:9.1f} {dev_total(e) / a.steps / 1e3:9.3f}")


if __name__ == "__main__":
    main()
