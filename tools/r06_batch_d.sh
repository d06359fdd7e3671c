#!/bin/bash
# round-6 batch D: the default bench line (traffic from the re-taken PMC file), then the whole -m gpu suite with its durations
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r06_bench_line_b16_v2.json 2> gpurun_out/r06_bench_line_b16_v2.err
python3 -c "
import json
l=[x for x in open('gpurun_out/r06_bench_line_b16_v2.json').read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['traffic'], d['parity_mode'].get('images_per_sec'))"
timeout 1700 python3 -m pytest tests -q -m gpu --durations=120 -p no:cacheprovider 2>&1 | grep -E "^[0-9.]+s (call|setup|teardown)|passed|failed" > gpurun_out/r06_gputest_f.txt
tail -3 gpurun_out/r06_gputest_f.txt
