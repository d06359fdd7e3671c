#!/bin/bash
# exact16 train step (the bench's parity mode as the main run) under two x2_plan values, alternated: tools/ab_x2plan6.sh 59 187 [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for plan in $A $B; do
    RESR_X2_PLAN=$plan timeout 400 python3 bench.py --precision exact16 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-other-configs --no-sustained 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'x2_plan':$plan,'round':$i,'images_per_s':d['value'],'ms_per_step':d['ms_per_step']}))"
  done
done
