"""Per-tensor gradient errors of the discriminator against the float64 oracle with conv1 scaled by --factor (activations of
O(factor)), per precision: which tensors lose accuracy when every activation is tiny."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import real_esrgan_pytorch_amd as R  # noqa: E402
from oracle import model_ref as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--factor", type=float, default=0.002)
ap.add_argument("--scale", type=float, default=256.0)
a = ap.parse_args()
sd = M.init_discriminator_state(11)
sd = {k: v.clone() for k, v in sd.items()}
sd["conv1.weight"] = sd["conv1.weight"] * a.factor
sd["conv1.bias"] = sd["conv1.bias"] * a.factor
gen = torch.Generator().manual_seed(2)
x = torch.rand(2, 3, 48, 64, generator=gen)
gw = torch.randn(2, 1, 48, 64, generator=gen)
sdo = {k: v.double().clone() for k, v in sd.items()}
for k in sdo:
    if not (k.endswith("_u") or k.endswith("_v")):
        sdo[k].requires_grad_(True)
xo = x.double().clone().requires_grad_(True)
yo = M.discriminator_forward(xo, sdo, True)
(yo * gw.double()).sum().backward()
for precision in ("strict", "exact16", "fast"):
    d = R.Discriminator(precision=precision)
    d.load_state_dict(sd)
    d = d.cuda().train()
    xd = x.cuda().requires_grad_(True)
    y = d(xd)
    sc = 1.0 if precision == "strict" else a.scale
    (y * gw.cuda()).sum().mul(sc).backward()
    torch.cuda.synchronize()
    rel = lambda p, q: ((p.double() - q).norm() / q.norm().clamp_min(1e-300)).item()
    errs = {n: rel(p.grad.cpu() / sc, sdo[n].grad) for n, p in d.named_parameters()}
    errs["x"] = rel(xd.grad.cpu() / sc, xo.grad)
    print(precision, "fwd", ((y.detach().cpu().double() - yo.detach()).abs().max()).item(), " ".join(f"{k}:{v:.1e}" for k, v in errs.items()))
