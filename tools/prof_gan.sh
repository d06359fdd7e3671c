#!/bin/bash
# rocprofv3 kernel trace of the GAN step (tools/bench_gan.py) -> gpurun_out/prof_gan/kernel_stats.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_gan
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_gan.py --steps 8 > $OUT/plain.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/bench_gan.py --steps 8 > $OUT/traced.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
if [ -n "$DB" ]; then python3 $R/tools/rocpd_summary.py $DB 45 > $OUT/kernel_stats.txt; python3 $R/tools/rocpd_gaps.py $DB >> $OUT/kernel_stats.txt; fi
rm -rf $OUT/trace
tail -1 $OUT/plain.json | cut -c1-200; tail -1 $OUT/traced.json | cut -c1-200
