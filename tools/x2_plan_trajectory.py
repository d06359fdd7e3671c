"""What exact16's backward plans mean for TRAINING: the same RealESRNet steps (fixed batches, no degradation draws: lr = area-downsampled
hr) from the same initial weights under
    exact16 plan 0 (pairs everywhere: the reference trajectory), plan 27 (default), plans 11 / 3 (without bit 4 / bits 3, 4), plan 31 (opt-in single store),
    exact16 plan 27 + hi-only weight gradients (RESR_X2_WGRAD_PRODUCTS=1), and fast (f16),
and after K Adam steps the distance of every trajectory's weights from plan 0's, relative to the distance plan 0 travelled from
the initial weights -- Adam divides by the gradient's own magnitude, so what counts is the DIRECTION error of the gradient, summed
over steps -- plus the loss curves.

    python tools/x2_plan_trajectory.py [--steps 40] [--batch 16] [--crop 64] [--out gpurun_out/x2_plan_trajectory.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import real_esrgan_pytorch_amd as R  # noqa: E402
from real_esrgan_pytorch_amd.train import RealESRNetStep  # noqa: E402


def batches(steps, n, crop, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    out = []
    for _ in range(steps):
        base = torch.rand(n, 3, crop // 4, crop // 4, device="cuda", generator=gen)
        hr = torch.nn.functional.interpolate(base, size=(4 * crop, 4 * crop), mode="bicubic").clamp(0, 1)
        hr = (0.9 * hr + 0.1 * torch.rand(n, 3, 4 * crop, 4 * crop, device="cuda", generator=gen)).clamp(0, 1)
        lr = torch.nn.functional.interpolate(hr, size=(crop, crop), mode="area")
        out.append((hr, lr))
    return out


def run(precision, plan, products, sd, data, lr_rate):
    os.environ.pop("RESR_X2_WGRAD_PRODUCTS", None)
    if products:
        os.environ["RESR_X2_WGRAD_PRODUCTS"] = str(products)
    g = R.Generator(3, 3, 4, precision=precision, x2_plan=plan).cuda()
    g.load_state_dict(sd)
    g.train()
    opt = torch.optim.Adam(g.parameters(), lr_rate, (0.9, 0.99))
    scaler = torch.amp.GradScaler("cuda")
    step = RealESRNetStep(g, None, opt, scaler)
    losses = []
    for hr, lr in data:
        losses.append(step(hr, lr))
    torch.cuda.synchronize()
    os.environ.pop("RESR_X2_WGRAD_PRODUCTS", None)
    return torch.cat([p.detach().double().flatten() for p in g.parameters()]).cpu(), [float(v) for v in losses], float(scaler.get_scale())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--crop", type=int, default=64)
    ap.add_argument("--lr", type=float, default=2e-4)     # the reference's (config.py model_lr)
    ap.add_argument("--out", default="gpurun_out/x2_plan_trajectory.json")
    a = ap.parse_args()
    torch.manual_seed(0)
    ref = R.Generator(3, 3, 4, precision="exact16", x2_plan=0).cuda()
    with torch.no_grad():
        ref.conv4.bias.add_(0.5)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    w0 = torch.cat([p.detach().double().flatten() for p in ref.parameters()]).cpu()
    data = batches(a.steps, a.batch, a.crop, 3)
    runs = {"exact16_plan0": ("exact16", 0, None), "exact16_plan0_again": ("exact16", 0, None), "exact16_plan27_default": ("exact16", 27, None),
            "exact16_plan11": ("exact16", 11, None), "exact16_plan3": ("exact16", 3, None), "exact16_plan31_single_store": ("exact16", 31, None),
            "exact16_plan27_hi_only_wgrad": ("exact16", 27, 1), "fast_f16": ("fast", 0, None),
            # round 6: the default plan with the MX backward-data stages, and exact16's forward in front of fast mode's backward pass
            "exact16_plan155_mx_backward": ("exact16", 27 + 128, None), "exact16_plan667_mx_backward_mx_wgrad": ("exact16", 27 + 128 + 512, None), "exact16_plan1691_mx_tail": ("exact16", 27 + 128 + 512 + 1024, None), "exact16_forward_f16_backward_plan256": ("exact16", 256, None)}
    if os.environ.get("TRAJ_ONLY"):
        keep = ["exact16_plan0"] + os.environ["TRAJ_ONLY"].split(",")
        runs = {k: v for k, v in runs.items() if k in keep}
    res = {}
    for name, (prec, plan, prod) in runs.items():
        res[name] = run(prec, plan, prod, sd, data, a.lr)
        print(name, "loss", [round(v, 5) for v in res[name][1][:3]], "...", [round(v, 5) for v in res[name][1][-3:]], flush=True)
    wr = res["exact16_plan0"][0]
    travelled = (wr - w0).norm().item()
    rep = {"steps": a.steps, "batch": a.batch, "crop": a.crop, "lr": a.lr, "travelled_by_plan0": travelled, "weights_norm": w0.norm().item(), "runs": {}}
    for name, (w, losses, scale) in res.items():
        rep["runs"][name] = {"distance_from_plan0_over_travelled": (w - wr).norm().item() / travelled,
                             "max_abs_weight_difference": (w - wr).abs().max().item(), "loss_first": losses[0], "loss_last": losses[-1],
                             "max_abs_loss_difference": max(abs(x - y) for x, y in zip(losses, res["exact16_plan0"][1])), "final_loss_scale": scale}
    print(json.dumps(rep, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(rep, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
