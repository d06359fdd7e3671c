#!/bin/bash
# round-6 batch F: the default bench line on the final sources + committed PMC file, then smoke() and the whole -m gpu suite
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r06_bench_line_b16_v6.json 2> gpurun_out/r06_bench_line_b16_v6.err
python3 -c "
import json
l=[x for x in open('gpurun_out/r06_bench_line_b16_v6.json').read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['traffic'], d['parity_mode'].get('value'))"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1700 python3 -m pytest tests -q -m gpu --durations=15 -p no:cacheprovider 2>&1 | grep -E "^[0-9.]+s (call|setup|teardown)|passed|failed|Error" > gpurun_out/r06_gputest_j.txt
tail -4 gpurun_out/r06_gputest_j.txt
