#!/bin/bash
# round-6 batch E (after the MX job took over g_lo's bias sums): profiles on the final sources, the bench line, the five-geometry gradient gate
bash tools/profile_round.sh r06c > /dev/null 2>&1
R=gpurun_out/prof_r06c
python3 bench.py > gpurun_out/r06_bench_line_b16_v3.json 2> gpurun_out/r06_bench_line_b16_v3.err
python3 -c "
import json
l=[x for x in open('gpurun_out/r06_bench_line_b16_v3.json').read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['traffic'], d['parity_mode'].get('value'))"
timeout 900 python3 tools/x2_plan_validate.py --plans 667 --seeds 5,6 --out gpurun_out/r06_x2_plan_validate_v2.json 2>&1 | grep -v "^{" | grep -A5 "plan667_.*dense" | grep -E "plan667|worst_rel|bias_worst" | head -40
ls $R
