"""Seam error of tiled inference against a whole-image pass, as a function of the halo.

A 23-block x4 generator on a 512 x 512 LR frame (the whole frame fits one launch sequence), stitched from 2 x 2 tiles of 256^2
with a halo of 16 / 32 / 64 / 128 LR pixels: max / mean absolute difference of the stitched SR image to the whole-image SR image,
and how many uint8 output values differ (the entry points write truncated uint8, imgproc.py:1594).  Reference weights at their
init scale and with the dense-block weights x 4 (a trained network's branches are not small); exact16 and fast.

Round 6: the curve is extended until the seam is VISIBLE -- dense weights x 8 / x 12 and conv1 x 40, halo 8 .. 128,
exact16 on the all-pairs plan (the lowest floor) -- so that the default halo is read off a curve: `halo_for_1e-4` per weight scale.

    python tools/tile_halo_error.py [--json profiles/r06_tile_halo_error.json]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import real_esrgan_pytorch_amd as R  # noqa: E402
from real_esrgan_pytorch_amd.tiling import TiledGenerator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--tile", type=int, default=256)
    a = ap.parse_args()
    torch.manual_seed(0)
    base = R.Generator(3, 3, 4, precision="exact16")
    with torch.no_grad():
        base.conv4.bias.add_(0.5)
    sd0 = {k: v.clone() for k, v in base.state_dict().items()}
    gen = torch.Generator().manual_seed(3)
    # image-like frame: smooth structure + grain (flat noise would make every tile look alike)
    x = F.interpolate(torch.rand(1, 3, a.size // 16, a.size // 16, generator=gen), size=(a.size, a.size), mode="bicubic").clamp(0, 1)
    x = (0.85 * x + 0.15 * torch.rand(1, 3, a.size, a.size, generator=gen)).clamp(0, 1).cuda()
    rows = []
    for wscale, c1scale in ((1.0, 1.0), (4.0, 1.0), (8.0, 1.0), (12.0, 1.0), (4.0, 40.0)):
        sd = {k: (v * wscale if k.endswith(".weight") and ".rdb" in k else v) for k, v in sd0.items()}
        if c1scale != 1.0:
            sd["conv1.weight"] = sd["conv1.weight"] * c1scale
        for precision in ("exact16", "fast"):
            g = R.Generator(3, 3, 4, precision=precision, x2_plan=0)
            g.load_state_dict(sd)
            g = g.cuda().eval()
            with torch.no_grad():
                whole = g(x)
            if not torch.isfinite(whole).all():
                print(json.dumps({"dense_weight_scale": wscale, "conv1_scale": c1scale, "precision": precision, "skipped": "non-finite output"}), flush=True)
                del g
                torch.cuda.empty_cache()
                continue
            for halo in (8, 16, 32, 64, 128):
                tiled = TiledGenerator(g, tile=a.tile, halo=halo, use_graph=False)(x)
                d = (tiled - whole).abs()
                u8 = ((tiled * 255).clamp(0, 255).to(torch.uint8) != (whole * 255).clamp(0, 255).to(torch.uint8)).float().mean().item()
                tiles, wh, ww = TiledGenerator(g, tile=a.tile, halo=halo, use_graph=False).plan(1, a.size, a.size)
                row = {"dense_weight_scale": wscale, "conv1_scale": c1scale, "precision": precision, "halo": halo, "max_abs": d.max().item(), "mean_abs": d.mean().item(),
                       "uint8_values_differing": u8, "computed_over_frame_pixels": round(len(tiles) * wh * ww / (a.size * a.size), 3),
                       "unclamped": ((whole > 0) & (whole < 1)).float().mean().item()}
                rows.append(row)
                print(json.dumps(row), flush=True)
            del g
            torch.cuda.empty_cache()
    # the smallest measured halo whose seam error is <= 1e-4 (max abs), per (weight scale, precision)
    need = {}
    for r in rows:
        key = f"dense x {r['dense_weight_scale']:g}, conv1 x {r['conv1_scale']:g}, {r['precision']}"
        if r["max_abs"] <= 1e-4 and key not in need:
            need[key] = r["halo"]
        need.setdefault(key + " (max_abs at halo 64)", None)
        if r["halo"] == 64:
            need[key + " (max_abs at halo 64)"] = r["max_abs"]
    print(json.dumps(need, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"frame": [a.size, a.size], "tile": a.tile, "model": "x4, 23 blocks, reference init (seed 0), conv4.bias + 0.5; exact16 on the all-pairs plan",
                       "halo_for_1e-4": need, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
