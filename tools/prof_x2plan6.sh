#!/bin/bash
# per-kernel totals of the exact16 train step under two x2_plan values (rocprofv3 --kernel-trace --stats): tools/prof_x2plan6.sh 187 699
HERE=$PWD; cd /tmp; export TMPDIR=/tmp
for plan in "$@"; do
  rm -rf /tmp/px_$plan
  RESR_X2_PLAN=$plan rocprofv3 --kernel-trace --stats -d /tmp/px_$plan -o t --output-format csv -- python3 $HERE/bench.py --precision exact16 --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-other-configs --no-sustained --no-probe > /dev/null 2>&1
  echo "== plan $plan"
  python3 - $plan <<'EOF2'
import csv, glob, sys
f = glob.glob(f"/tmp/px_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%6d x %9.1f us  %7.1f ms %5.1f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:100]))
print("total %.1f ms" % (tot / 1e6))
EOF2
done
