"""exact16's backward plans against the all-pairs plan ON THE GPU, at the geometries training runs at (the emulation of
tools/precision_ladder_sim.py stops at 1 x 128^2: a float64 evaluation of 16 x 256^2 is hours of CPU).

The plans share their forward pass bit for bit (hence every LeakyReLU mask), so the distance between their gradients is the
rung's own effect, free of mask flips: per-tensor relative L2 of all 702 gradient tensors (23 blocks) for
  plan 3 (default: growth-plane gradients stored as pairs, read as their hi tensor, bias sums from hi + lo) and
  plan 7 (opt-in: stored single)
against plan 0, under the L1 loss of the train step on image-like input and under a dense random cotangent.

    python tools/x2_plan_validate.py [--out gpurun_out/x2_plan_validate.json] [--cases 16x256,2x256,...] [--seeds 5,6,7]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import real_esrgan_pytorch_amd as R  # noqa: E402


def grads(g, plan, x, loss_of, scale=1024.0, products=None):
    if g.x2_plan != plan:      # (the MX plans have their own workspaces -- 80-100 GB each at 16 x 256^2: one at a time)
        g._workspaces.clear()
        torch.cuda.empty_cache()
    g.x2_plan = plan
    os.environ.pop("RESR_X2_WGRAD_PRODUCTS", None)
    if products:
        os.environ["RESR_X2_WGRAD_PRODUCTS"] = str(products)
    g.zero_grad(set_to_none=True)
    y = g(x)
    loss_of(y).mul(scale).backward()
    torch.cuda.synchronize()
    return y.detach().clone(), {n: p.grad.detach().double().cpu() / scale for n, p in g.named_parameters()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/x2_plan_validate.json")
    ap.add_argument("--cases", default="16x256,2x256,32x64,1x128,1x24")
    ap.add_argument("--seeds", default="5,6,7")
    ap.add_argument("--l1-scales", default="1024", help="loss scales of the L1 runs (a GradScaler starts at 65536 and doubles every 2000 clean steps)")
    ap.add_argument("--prescale-targets", default="", help="e.g. 0,3,6,9,12: the dense-cotangent runs at amplitude 2^-20 (always lifted) "
                                                            "under each $RESR_X2_GRAD_PRESCALE_LOG2 -- how high must the lift go")
    ap.add_argument("--plans", default="3,7", help="plans compared with plan 0 (11 = 3 + the weight products read the growth planes as hi)")
    ap.add_argument("--hi-only", action="store_true", help="also the opt-in hi-only weight gradients (RESR_X2_WGRAD_PRODUCTS=1) on top of plan 3")
    a = ap.parse_args()
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="exact16").cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    rep = {}
    for case in a.cases.split(","):
        n, s = (int(v) for v in case.split("x"))
        for seed in (int(v) for v in a.seeds.split(",")):
            gen = torch.Generator(device="cuda").manual_seed(seed)
            # image-like input: smooth field + grain (the regime training runs in)
            x = torch.nn.functional.interpolate(torch.rand(n, 3, s // 8, s // 8, device="cuda", generator=gen), size=(s, s), mode="bicubic").clamp(0, 1)
            x = (0.9 * x + 0.1 * torch.rand(n, 3, s, s, device="cuda", generator=gen)).clamp(0, 1)
            target = torch.rand(n, 3, 4 * s, 4 * s, device="cuda", generator=gen)
            gw = torch.randn(n, 3, 4 * s, 4 * s, device="cuda", generator=gen) / (4 * s)
            runs = [(f"l1_x{int(sc)}", lambda y: (y - target).abs().mean(), sc) for sc in (float(v) for v in a.l1_scales.split(","))]
            runs.append(("dense", lambda y: (y * gw).sum(), 1024.0))
            for t in (int(v) for v in a.prescale_targets.split(",") if v):
                runs.append((f"dense_tiny_target{t}", lambda y: (y * gw).sum(), 2.0 ** -20))
            for lname, loss_of, sc in runs:
                os.environ.pop("RESR_X2_GRAD_PRESCALE_LOG2", None)
                if "_target" in lname:
                    os.environ["RESR_X2_GRAD_PRESCALE_LOG2"] = lname.split("_target")[1]
                y0, g0 = grads(g, 0, x, loss_of, sc)
                row = {}
                for plan in [int(v) for v in a.plans.split(",")] + ["3_hi_only_wgrad"]:
                    if plan == "3_hi_only_wgrad":
                        if not a.hi_only:
                            continue
                        yp, gp = grads(g, 3, x, loss_of, sc, products=1)
                    else:
                        yp, gp = grads(g, plan, x, loss_of, sc)
                    rel = {k: ((gp[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30)).item() for k in g0}
                    worst = max(rel, key=rel.get)
                    vals = sorted(rel.values())
                    row[f"plan{plan}"] = {"forward_equal": bool(torch.equal(yp, y0)), "worst_tensor": worst, "worst_rel_l2": rel[worst],
                                          "median_rel_l2": vals[len(vals) // 2], "p99_rel_l2": vals[int(len(vals) * 0.99)], "tensors": len(vals),
                                          "weights_worst": max(v for k, v in rel.items() if k.endswith("weight")),
                                          "bias_worst": max(v for k, v in rel.items() if k.endswith("bias"))}
                rep[f"{case}_s{seed}_{lname}"] = row
                print(case, seed, lname, json.dumps(row), flush=True)
    summary = {}
    lnames = sorted({k.split("_", 2)[2] for k in rep})
    for plan in tuple(f"plan{v}" for v in a.plans.split(",")) + (("plan3_hi_only_wgrad",) if a.hi_only else ()):
        for case in a.cases.split(","):
          for ln in lnames:
            rows = [v[plan] for k, v in rep.items() if k.startswith(case + "_") and k.endswith("_" + ln)]
            summary[f"{plan}_{case}_{ln}"] = {"worst_rel_l2": max(r["worst_rel_l2"] for r in rows), "weights_worst": max(r["weights_worst"] for r in rows),
                                         "bias_worst": max(r["bias_worst"] for r in rows), "median_max": max(r["median_rel_l2"] for r in rows), "runs": len(rows)}
    rep["summary"] = summary
    print(json.dumps(summary, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(rep, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
