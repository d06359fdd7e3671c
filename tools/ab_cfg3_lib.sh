#!/bin/bash
# config 3 (RealESRNet B = 32, HR 256^2) and the GAN step under two builds of the library, alternated: tools/ab_cfg3.sh tools/ab/A.so tools/ab/B.so [rounds]
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for lib in $A $B; do
    RESR_LIB_PATH=$PWD/$lib STEPS=80 timeout 200 python3 tools/bench_cfg3.py 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'lib':'$lib','round':$i,'what':'cfg3','images_per_s':d['images_per_s'],'ms_per_step':d['ms_per_step']}))"
    RESR_LIB_PATH=$PWD/$lib STEPS=40 timeout 200 python3 tools/bench_gan.py 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'lib':'$lib','round':$i,'what':'gan','ms_per_step':d['ms_per_step']}))"
  done
done
