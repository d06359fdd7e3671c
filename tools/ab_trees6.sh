#!/bin/bash
# Same-box A/B of two whole trees (library + Python), alternated: headline step, config 3, GAN step, config 5.
#   tools/ab_trees6.sh <other tree> [rounds] [out.json]
# Every measurement is its own process (fresh clocks, fresh allocator); the other tree runs with ITS OWN tools/.
O=$1; R=${2:-2}; OUT=${3:-gpurun_out/ab_trees6.jsonl}; HERE=$PWD
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
pick() { python3 -c "
import json,sys
which,i,name,keys=sys.argv[1],sys.argv[2],sys.argv[3],sys.argv[4].split(',')
last=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]
want=[l for l in last if sys.argv[5] in l] if len(sys.argv)>5 else last
d=json.loads(want[-1]) if want else {}
r={'tree':which,'round':int(i),'what':name}
for k in keys:
    v=d
    for p in k.split('.'):
        v=v.get(p) if isinstance(v,dict) else None
    r[k]=v
print(json.dumps(r))" "$@"; }
for i in $(seq 1 $R); do
  for which in head other; do
    if [ $which = other ]; then cd $HERE/$O; else cd $HERE; fi
    timeout 300 python3 bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --steps 16 --warmup 4 2>/dev/null | pick $which $i headline value,ms_per_step,roofline.vs_sustained.matrix_alone_tflops >> $HERE/$OUT
    STEPS=60 timeout 200 python3 tools/bench_cfg3.py 2>/dev/null | pick $which $i cfg3 images_per_s,ms_per_step >> $HERE/$OUT
    timeout 200 python3 tools/bench_gan.py --steps 16 2>/dev/null | pick $which $i gan ms_per_step >> $HERE/$OUT
    INF=$(timeout 300 python3 tools/bench_infer.py 2>/dev/null)
    echo "$INF" | pick $which $i cfg5 frames_per_s,ms "\"config\": \"5" >> $HERE/$OUT
    echo "$INF" | pick $which $i cfg2 images_per_s,ms "\"config\": \"2" >> $HERE/$OUT
  done
done
cd $HERE; cat $OUT
