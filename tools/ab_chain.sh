#!/bin/bash
# same-box A/B of the chained dense-block launches (RESR_CONV_NO_CHAIN=1 = one launch per pass): config 3, GAN step, headline
R=${1:-2}
for i in $(seq 1 $R); do
  for which in chain nochain; do
    if [ $which = nochain ]; then export RESR_CONV_NO_CHAIN=1; else unset RESR_CONV_NO_CHAIN; fi
    echo -n "$which $i cfg3: "; STEPS=20 timeout 200 python tools/bench_cfg3.py 2>/dev/null | tail -1 | cut -c1-140
    echo -n "$which $i gan: "; timeout 200 python tools/bench_gan.py 2>/dev/null | tail -1 | cut -c1-140
    echo -n "$which $i headline: "; timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pi=d['roofline']['per_instance']
print(d['value'], d['ms_per_step'], {k.split('<')[1][:-1] if '<' in k else k:(round(v['tflops']),round(v['ms_per_step'],2),v['launches']) for k,v in pi.items()})"
  done
done
