#!/bin/bash
# same-box A/B of two whole trees (library + Python): tools/ab_trees.sh <other tree> [rounds] -- headline, config 3, GAN step
O=$1; R=${2:-2}; HERE=$PWD
for i in $(seq 1 $R); do
  for which in tree other; do
    if [ $which = other ]; then cd $HERE/$O; else cd $HERE; fi
    echo -n "$which $i headline: "; timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
    echo -n "$which $i cfg3: "; STEPS=40 timeout 200 python tools/bench_cfg3.py 2>/dev/null | tail -1 | cut -c78-120
    echo -n "$which $i gan: "; timeout 200 python tools/bench_gan.py --steps 12 2>/dev/null | tail -1 | cut -c88-125
  done
done
