#!/bin/bash
# round-6 batch B: the default plan's gradients against the all-pairs plan at the five training geometries (VERDICT round 5, item 1's gate),
# and the chained launches' soak (fast mode config 3, 200 steps; exact16 default plan, 30 steps at 16 x 64^2 through PRECISION)
timeout 1500 python3 tools/x2_plan_validate.py --plans 27,155,667 --seeds 5,6 --out gpurun_out/r06_x2_plan_validate.json 2>&1 | grep -v "^{" | tail -70
timeout 600 python3 tools/chain_soak.py --steps 200 2>&1 | tail -3
# experiment: the MX weight-gradient job's X fragments read once per halo row and shifted in registers (RESR_WGRAD_MX_SHIFT)
if [ -f tools/ab/mxshift.so ]; then
  RESR_LIB_PATH=$PWD/tools/ab/mxshift.so timeout 600 python3 -m pytest tests/test_gpu_mx.py -q -k "weight or wgrad" -p no:cacheprovider 2>&1 | tail -4
  bash tools/ab_lib6.sh tools/ab/base6.so tools/ab/mxshift.so 2 | tee gpurun_out/r06_ab_mxshift.jsonl
fi
