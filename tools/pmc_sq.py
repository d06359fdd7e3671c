"""Reduce one rocprofv3 --pmc pass of SQ counters (csv) to per-kernel averages and the derived shares the guide names
(MI355X_MICROARCH.md, rocprofv3 PMC slots): WAIT_ANY (wave parked on s_waitcnt / barrier), WAIT_INST_ANY (issue stall),
ACTIVE_INST_ANY -- disjoint, summing to ~WAVE_CYCLES (quad-cycles); MFMA busy cycles per wave cycle."""
import collections
import csv
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import short  # noqa: E402

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0 or not any(s in k for s in ("conv3x3", "wgrad")):
        continue
    rec = {"launches": len(next(iter(cs.values()))), **{c: round(v) for c, v in m.items()}}
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
        if c in m:
            rec[c.lower() + "_share"] = round(m[c] / wc, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        rec["mfma_busy_cycles_per_wave_quadcycle"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / wc, 4)
    out[k] = rec
print(json.dumps(out, indent=1))
