#!/bin/bash
# same-box A/B of BASELINE config 3 (and the GAN step with GAN=1): tools/ab_cfg3.sh <variant.so> [rounds]
V=$1; R=${2:-2}
for i in $(seq 1 $R); do
  for which in tree variant; do
    if [ $which = variant ]; then export RESR_LIB_PATH=$PWD/$V; else unset RESR_LIB_PATH; fi
    echo -n "$which $i cfg3: "; STEPS=20 timeout 200 python tools/bench_cfg3.py 2>/dev/null | tail -1
    if [ -n "$GAN" ]; then echo -n "$which $i gan: "; timeout 200 python tools/bench_gan.py 2>/dev/null | tail -1; fi
  done
done
