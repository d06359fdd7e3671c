#!/bin/bash
# Same-box A/B of ONE environment switch on the headline step (and config 3), alternated: tools/ab_env6.sh VAR=VALUE [rounds] [out.jsonl]
KV=$1; R=${2:-2}; OUT=${3:-gpurun_out/ab_env6.jsonl}; mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for i in $(seq 1 $R); do
  for which in default switched; do
    if [ $which = switched ]; then export "$KV"; else unset "${KV%%=*}"; fi
    timeout 300 python3 bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --steps 16 --warmup 4 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'arm':'$which','env':'$KV','round':$i,'what':'headline','value':d['value'],'ms_per_step':d['ms_per_step'],'power_w':d.get('power',{}).get('mean_w')}))" >> $OUT
    STEPS=60 timeout 200 python3 tools/bench_cfg3.py 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'arm':'$which','env':'$KV','round':$i,'what':'cfg3','images_per_s':d['images_per_s'],'ms_per_step':d['ms_per_step']}))" >> $OUT
  done
done
unset "${KV%%=*}"; cat $OUT
