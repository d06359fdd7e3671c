#!/bin/bash
# SQ counters of the degradation stage's kernels alone (tools/time_degrade.py under rocprofv3 --pmc; no tracing): bash tools/pmc_degrade.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_degrade; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
   --output-format csv -d $OUT/sq -o p -- python3 $R/tools/time_degrade.py > /dev/null 2> $OUT/err.txt
S=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "resr" not in k and "usm" not in k: continue
    wc = sum(v["SQ_WAVE_CYCLES"]) or 1
    print(k, len(v["SQ_WAVE_CYCLES"]), {c: round(sum(x) / wc, 3) for c, x in v.items() if c != "SQ_WAVE_CYCLES"})
PY
rm -rf $OUT/sq
