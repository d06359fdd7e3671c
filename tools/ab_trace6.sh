#!/bin/bash
# Per-kernel totals of the headline step (and config 3) under rocprofv3 --kernel-trace --stats for two trees on one box:
#   tools/ab_trace6.sh <other tree>   ->  gpurun_out/ab_trace6/{head,other}_{headline,cfg3}_stats.csv
O=$1; HERE=$PWD; OUT=$HERE/gpurun_out/ab_trace6; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for which in head other; do
  if [ $which = other ]; then T=$HERE/$O; else T=$HERE; fi
  cd $T
  rm -rf /tmp/abt_$which; rocprofv3 --kernel-trace --stats -d /tmp/abt_$which -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --no-sustained --no-probe --steps 8 --warmup 3 > $OUT/${which}_headline.log 2>&1
  cp $(find /tmp/abt_$which -name '*kernel_stats.csv' | head -1) $OUT/${which}_headline_stats.csv
  rm -rf /tmp/abt3_$which; STEPS=30 rocprofv3 --kernel-trace --stats -d /tmp/abt3_$which -o t --output-format csv -- python3 tools/bench_cfg3.py > $OUT/${which}_cfg3.log 2>&1
  cp $(find /tmp/abt3_$which -name '*kernel_stats.csv' | head -1) $OUT/${which}_cfg3_stats.csv
done
ls -la $OUT
