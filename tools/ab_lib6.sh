#!/bin/bash
# exact16 train step (the bench's parity mode as the main run) under two builds of the library, alternated: tools/ab_lib6.sh tools/ab/base6.so tools/ab/exp.so [rounds] [extra bench flags]
A=$1; B=$2; R=${3:-2}; shift 3
for i in $(seq 1 $R); do
  for lib in $A $B; do
    RESR_LIB_PATH=$PWD/$lib timeout 400 python3 bench.py --precision exact16 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-other-configs --no-sustained "$@" 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'lib':'$lib','round':$i,'images_per_s':d['value'],'ms_per_step':d['ms_per_step']}))"
  done
done
