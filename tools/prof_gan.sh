#!/bin/bash
# rocprofv3 kernel trace of the GAN step (tools/bench_gan.py --content = BASELINE config 4's per-GPU share) -> gpurun_out/prof_gan/
#   kernel_stats.txt  = STEADY-STATE steps only (tools/rocpd_steady.py: the last 4 whole steps, per-step figures)
# Usage: tools/prof_gan.sh [tag]   (env knobs such as RESR_UNFUSED_LOSSES=1 / RESR_PER_TENSOR_ADAM=1 pass through)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-gan}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_gan.py --content --steps 10 > $OUT/plain.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/bench_gan.py --content --steps 10 > $OUT/traced.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
if [ -n "$DB" ]; then python3 $R/tools/rocpd_steady.py $DB 4 60 > $OUT/kernel_stats.txt; python3 $R/tools/rocpd_steady.py $DB --list "*" > $OUT/last_step_launches.txt; fi
rm -rf $OUT/trace
tail -1 $OUT/plain.json | cut -c1-200; tail -1 $OUT/traced.json | cut -c1-200
