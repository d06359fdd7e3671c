#!/bin/bash
# config 3 under several builds of the library, in turn, twice: tools/ab_cfg3n.sh tools/ab/A.so tools/ab/B.so ...
for i in 1 2; do
  for lib in "$@"; do
    RESR_LIB_PATH=$PWD/$lib STEPS=80 timeout 200 python3 tools/bench_cfg3.py 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'lib':'$lib','round':$i,'images_per_s':d['images_per_s'],'ms_per_step':d['ms_per_step'],'loss':d.get('loss')}))"
  done
done
