#!/bin/bash
# round-6 batch C: the default bench line on the final sources, and where the plain-launch test's wall time goes
mkdir -p gpurun_out
( time python3 bench.py > gpurun_out/r06_bench_line_b16_v2.json 2> gpurun_out/r06_bench_line_b16_v2.err ) 2>&1 | tail -4
tail -c 600 gpurun_out/r06_bench_line_b16_v2.json; echo
S=$(date +%s.%N)
RESR_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 2 --warmup 1 --batch 2 --lr-size 32 --gan --no-parity-mode --no-sustained 2>&1 | while IFS= read -r l; do printf '%7.2f %s\n' "$(echo "$(date +%s.%N) - $S" | bc)" "${l:0:160}"; done | tail -40
