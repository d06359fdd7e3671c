#!/bin/bash
# the headline step under two builds of the library, alternated: tools/ab_head.sh tools/ab/A.so tools/ab/B.so [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for lib in $A $B; do
    RESR_LIB_PATH=$PWD/$lib timeout 300 python3 bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --steps 16 --warmup 4 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'lib':'$lib','round':$i,'value':d['value'],'ms_per_step':d['ms_per_step']}))"
  done
done
