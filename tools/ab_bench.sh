#!/bin/bash
# same-box A/B of the headline step: tools/ab_bench.sh <variant.so> [rounds] -- alternates the in-tree build and the variant
V=$1; R=${2:-2}; shift; shift
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for which in tree variant; do
    if [ $which = variant ]; then export RESR_LIB_PATH=$PWD/$V; else unset RESR_LIB_PATH; fi
    timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --steps 12 --warmup 4 "$@" 2>/dev/null | tail -1 > gpurun_out/ab_${which}_$i.json
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_${which}_$i.json"))
pi=d["roofline"]["per_instance"]
print("$which $i", d["value"], d["ms_per_step"], {k.split("<")[1][:-1] if "<" in k else k:(round(v["tflops"]),round(v["ms_per_step"],2)) for k,v in pi.items()})
PY
  done
done
