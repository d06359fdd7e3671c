#!/bin/bash
# Upper bound of a Winograd F(2x2,3x3) formulation of the fast-mode convolutions, measured on this kernel's own memory side
# (DESIGN section 7): the in-tree build against a timing variant that issues 4 of 9 tap products (tools/build_variant.py wino4
# -DRESR_TIMING_TAPS=4; wrong results by design).  Back-to-back launches of the dense-block shapes (tools/conv_loop.py) and the
# headline train step, alternated on one box.   bash tools/winograd_bound.sh > gpurun_out/winograd_bound.txt
# wino4v additionally charges the consumer waves 8 packed-f16 vector instructions per remaining MFMA (-DRESR_TIMING_VALU=8): the
# input transform's share if it runs in registers.
for which in tree wino4 wino4v; do
  if [ $which != tree ]; then export RESR_LIB_PATH=$PWD/tools/ab/$which.so; else unset RESR_LIB_PATH; fi
  echo "== conv_loop, $which"
  python tools/conv_loop.py --shapes 64:32,160:32,192:64 2>/dev/null
done
for i in 1 2; do
  for which in tree wino4 wino4v; do
    if [ $which != tree ]; then export RESR_LIB_PATH=$PWD/tools/ab/$which.so; else unset RESR_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --steps 12 --warmup 4 2>/dev/null | tail -1 > gpurun_out/wb_${which}_$i.json
    python - <<PY
import json
d=json.load(open("gpurun_out/wb_${which}_$i.json"))
pi=d["roofline"]["per_instance"]
pw=d.get("power") or {}
print("== step, $which $i:", d["value"], "images/s", d["ms_per_step"], "ms;", pw.get("avg_w"), "W", pw.get("sclk_mhz_avg"), "MHz;",
      {k.split("<")[1][:-1] if "<" in k else k: (round(v["ms_per_step"], 2), "ms") for k, v in pi.items()})
PY
  done
done
