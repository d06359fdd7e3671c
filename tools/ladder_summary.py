"""Merge the JSON files of tools/ladder_round5.sh into ONE record (profiles/r05_precision_ladder_sim.json) and print the
per-geometry range table of DESIGN.md section 2.

    python tools/ladder_summary.py gpurun_out/sim profiles/r05_precision_ladder_sim.json
"""
import glob
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {}
for path in sorted(glob.glob(os.path.join(src, "r5_*x*_w*.json")) + glob.glob(os.path.join(src, "r5_bf8_*.json")) + glob.glob(os.path.join(src, "r5t2_*x*_w*.json")) + glob.glob(os.path.join(src, "r5t4_*x*_w*.json")) + glob.glob(os.path.join(src, "r5t5_*x*_w*.json")) + glob.glob(os.path.join(src, "r5i40_*x*_w*.json")) + glob.glob(os.path.join(src, "r5t6_*x*_w*.json"))):
    base = os.path.basename(path)
    tag = base[5:-5] if base.startswith(("r5t2_", "r5t4_", "r5t5_", "r5t6_")) else base[6:-5] if base.startswith("r5i40_") else base[3:-5]      # r5t2_* / r5t4_*: the TRAIN2 / TRAIN4 rungs, run separately, merged into their geometry
    for seed, rows in json.load(open(path)).items():
        out.setdefault(tag, {}).setdefault(seed, {}).update(rows)
for path in sorted(glob.glob(os.path.join(src, "r5l1_*x*_w*.json"))):   # --loss l1: the train step's own loss, kept as separate geometries
    tag = os.path.basename(path)[5:-5] + "_l1loss"
    for seed, rows in json.load(open(path)).items():
        out.setdefault(tag, {}).setdefault(seed, {}).update({name + " [L1 loss]": r for name, r in rows.items()})
json.dump({"tool": "tools/precision_ladder_sim.py --set round5 (23 blocks, float64 emulation of the storage roundings)", "runs": out}, open(dst, "w"), indent=1)
rungs = {}
for tag, seeds in out.items():
    for seed, rows in seeds.items():
        for name, r in rows.items():
            rungs.setdefault(name, {}).setdefault(tag, []).append(r)
tags = sorted(out)
print("| rung | " + " | ".join(tags) + " |")
print("|---|" + "---|" * len(tags))
for name, by in rungs.items():
    cells = []
    for t in tags:
        rs = by.get(t)
        if not rs:
            cells.append("")
            continue
        gw = [r["grad_worst"] for r in rs]
        fw = [r["fwd_max_abs"] for r in rs]
        cells.append(f"{min(gw):.1e}-{max(gw):.1e} (fwd {max(fw):.1e}, {len(rs)} seeds)")
    print(f"| {name} | " + " | ".join(cells) + " |")
