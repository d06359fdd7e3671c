#!/bin/bash
# rocprofv3 kernel trace of BASELINE config 3 (tools/bench_cfg3.py) -> gpurun_out/prof_cfg3/: kernel_stats.txt = STEADY-STATE steps
# only (tools/rocpd_steady.py: the last 4 whole steps, per-step figures), last_step_launches.txt = every launch of the last step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_cfg3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_cfg3.py > $OUT/plain.json 2>/dev/null
STEPS=10 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/bench_cfg3.py > $OUT/traced.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
if [ -n "$DB" ]; then python3 $R/tools/rocpd_steady.py $DB 4 50 > $OUT/kernel_stats.txt; python3 $R/tools/rocpd_steady.py $DB --list "*" > $OUT/last_step_launches.txt; fi
rm -rf $OUT/trace
tail -1 $OUT/plain.json | cut -c1-200; tail -1 $OUT/traced.json | cut -c1-200
