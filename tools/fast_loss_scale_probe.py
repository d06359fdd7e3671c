"""Fast mode's (f16) gradient error against exact16's all-pairs plan under the L1 mean loss, as a function of the loss scale:
how much of fast mode's gradient error at a GradScaler's initial scale is f16 UNDERFLOW (curable by a power-of-two lift, as
exact16's backward pass has) and how much is its 11-bit arithmetic.

    python tools/fast_loss_scale_probe.py [--cases 16x256,32x64] [--scales 16,20,24,28]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import real_esrgan_pytorch_amd as R  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="16x256,32x64")
ap.add_argument("--scales", default="10,16,20,24,28")
ap.add_argument("--out", default="gpurun_out/fast_loss_scale_probe.json")
a = ap.parse_args()
torch.manual_seed(0)
ref = R.Generator(3, 3, 4, precision="exact16", x2_plan=0).cuda().train()
with torch.no_grad():
    ref.conv4.bias.add_(0.5)
fast = R.Generator(3, 3, 4, precision="fast").cuda().train()
fast.load_state_dict(ref.state_dict())


def grads(g, x, target, scale):
    g.zero_grad(set_to_none=True)
    y = g(x)
    ((y - target).abs().mean() * scale).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().double().cpu() / scale for n, p in g.named_parameters()}


rep = {}
for case in a.cases.split(","):
    n, s = (int(v) for v in case.split("x"))
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.nn.functional.interpolate(torch.rand(n, 3, s // 8, s // 8, device="cuda", generator=gen), size=(s, s), mode="bicubic").clamp(0, 1)
    x = (0.9 * x + 0.1 * torch.rand(n, 3, s, s, device="cuda", generator=gen)).clamp(0, 1)
    target = torch.rand(n, 3, 4 * s, 4 * s, device="cuda", generator=gen)
    g0 = grads(ref, x, target, 1.0)
    for e in (int(v) for v in a.scales.split(",")):
        gf = grads(fast, x, target, 2.0 ** e)
        rel = sorted(((gf[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-300)).item() for k in g0)
        nonfinite = sum(int(not torch.isfinite(v).all()) for v in gf.values())
        rep[f"{case}_2^{e}"] = {"median": rel[len(rel) // 2], "p90": rel[int(len(rel) * 0.9)], "worst": rel[-1], "non_finite_tensors": nonfinite}
        print(case, f"loss scale 2^{e}", json.dumps(rep[f"{case}_2^{e}"]), flush=True)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
json.dump(rep, open(a.out, "w"), indent=1)
