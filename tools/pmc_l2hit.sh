#!/bin/bash
# L2 hit rate (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum; its own pass, no tracing) of the generator's forward kernels at a small geometry:
#   tools/pmc_l2hit.sh <batch> <res>   ->  gpurun_out/pmc_l2hit/<batch>x<res>.txt  (per kernel: launches, hits, misses, hit rate, miss bytes per launch at 128 B)
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-32}; S=${2:-64}
OUT=$R/gpurun_out/pmc_l2hit
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/raw_${B}x${S} -o p -- python3 $R/tools/fwd_loop.py --batch $B --res $S --reps 4 --train > /dev/null 2> $OUT/${B}x${S}.err
F=$(find $OUT/raw_${B}x${S} -name "*counter_collection.csv" | head -1)
python3 - <<PY > $OUT/${B}x${S}.txt
import csv, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open("$F")):
    k = r["Kernel_Name"][:90]
    a = acc[k]
    if r["Counter_Name"] == "TCC_HIT_sum": a[1] += float(r["Counter_Value"]); a[0] += 1
    if r["Counter_Name"] == "TCC_MISS_sum": a[2] += float(r["Counter_Value"])
for k, (n, h, m) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
    if n: print(f"{k:90s} launches {n:5d}  hit {h / n:12.0f}  miss {m / n:12.0f}  hit rate {h / max(1.0, h + m):.3f}  miss MB/launch {m / n * 128 / 1e6:8.2f}")
PY
rm -rf $OUT/raw_${B}x${S}
head -12 $OUT/${B}x${S}.txt
