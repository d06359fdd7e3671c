#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_energy
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/energy.py --only wgrad --seconds 0.5 > $OUT/out.txt 2> $OUT/err.txt
DB=$(find $OUT/trace -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB 10 > $OUT/kernel_stats.txt
rm -rf $OUT/trace
cat $OUT/out.txt | tail -3; cat $OUT/kernel_stats.txt | cut -c1-200
