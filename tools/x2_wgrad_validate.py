"""exact16 weight gradients from the hi tensors only (RESR_X2_WGRAD_PRODUCTS=1) against the three-product form, on the
geometries training runs at: per-tensor relative L2 of all 702 gradient tensors (23 blocks), forward untouched.

    python tools/x2_wgrad_validate.py [--out gpurun_out/x2_wgrad_validate.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import real_esrgan_pytorch_amd as R  # noqa: E402


def grads(g, x, gw, products):
    os.environ["RESR_X2_WGRAD_PRODUCTS"] = str(products)
    g.zero_grad(set_to_none=True)
    y = g(x)
    (y * gw).sum().mul(1024.0).backward()
    torch.cuda.synchronize()
    return y.detach().clone(), {n: p.grad.detach().double().cpu() / 1024.0 for n, p in g.named_parameters()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/x2_wgrad_validate.json")
    a = ap.parse_args()
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="exact16").cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    rep = {}
    for name, (n, s) in {"2x256": (2, 256), "16x64": (16, 64), "1x24": (1, 24)}.items():
        gen = torch.Generator(device="cuda").manual_seed(5)
        # image-like input: smooth field + grain (the regime training runs in)
        x = torch.nn.functional.interpolate(torch.rand(n, 3, s // 8, s // 8, device="cuda", generator=gen), size=(s, s), mode="bicubic").clamp(0, 1)
        x = (0.9 * x + 0.1 * torch.rand(n, 3, s, s, device="cuda", generator=gen)).clamp(0, 1)
        gw = torch.randn(n, 3, 4 * s, 4 * s, device="cuda", generator=gen) / (4 * s)
        y3, g3 = grads(g, x, gw, 3)
        y1, g1 = grads(g, x, gw, 1)
        rel = {k: ((g1[k] - g3[k]).norm() / g3[k].norm().clamp_min(1e-30)).item() for k in g3}
        worst = max(rel, key=rel.get)
        vals = sorted(rel.values())
        rep[name] = {"forward_equal": bool(torch.equal(y1, y3)), "worst_tensor": worst, "worst_rel_l2": rel[worst],
                     "median_rel_l2": vals[len(vals) // 2], "p99_rel_l2": vals[int(len(vals) * 0.99)], "tensors": len(vals),
                     "weights_worst": max(v for k, v in rel.items() if k.endswith("weight")),
                     "bias_worst": max(v for k, v in rel.items() if k.endswith("bias"))}
        print(name, json.dumps(rep[name]))
    os.environ.pop("RESR_X2_WGRAD_PRODUCTS", None)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(rep, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
