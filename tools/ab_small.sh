#!/bin/bash
# same-box A/B of the small-geometry steps (config 3, GAN step): tools/ab_small.sh <variant.so> [rounds] -- alternates the in-tree build and the variant
V=$1; R=${2:-2}
for i in $(seq 1 $R); do
  for which in tree variant; do
    if [ $which = variant ]; then export RESR_LIB_PATH=$PWD/$V; else unset RESR_LIB_PATH; fi
    python bench.py --lr-size 64 --batch 32 --steps 40 --warmup 10 --no-cpu-baseline --no-parity-mode --no-other-configs --no-probe --centre-output 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $i cfg3:', d['value'], d['ms_per_step'], 'loss', d['loss'], 'unclamped', d['unclamped_output_fraction'])"
    python bench.py --gan --steps 30 --warmup 8 --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $i gan:', d['value'], d['ms_per_step'], 'chain_errors', d['chain_errors'])"
  done
done
