#!/bin/bash
# round-6 measurement batch A (one gpurun call): the WGX_PREP variant of the MX weight-gradient kernel against the in-tree one, the opt-in
# exact16-forward / f16-backward plan, the 40-step trajectories, the tiler's halo curve
run() { RESR_X2_PLAN=$1 timeout 400 python3 bench.py --precision exact16 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-other-configs --no-sustained 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps({'what':'$2','x2_plan':$1,'images_per_s':d['value'],'ms_per_step':d['ms_per_step']}))"; }
for i in 1 2; do
  run 763 in_tree
  RESR_LIB_PATH=$PWD/tools/ab/wgxprep.so run 763 wgx_prep_variant
done
run 256 exact16_forward_f16_backward
run 59 round5_default
TRAJ_ONLY=fast_f16,exact16_plan27_default,exact16_plan155_mx_backward,exact16_plan667_mx_backward_mx_wgrad,exact16_forward_f16_backward_plan256 timeout 900 python3 tools/x2_plan_trajectory.py --out gpurun_out/r06_x2_plan_trajectory.json 2>&1 | tail -8
timeout 900 python3 tools/tile_halo_error.py --json gpurun_out/r06_tile_halo_error.json 2>&1 | tail -14
