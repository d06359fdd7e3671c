#!/bin/bash
# rocprofv3 kernel trace of BASELINE config 3 (tools/bench_cfg3.py) -> gpurun_out/prof_cfg3/kernel_stats.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_cfg3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=8 python3 $R/tools/bench_cfg3.py > $OUT/plain.json 2>/dev/null
STEPS=8 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/tools/bench_cfg3.py > $OUT/traced.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
if [ -n "$DB" ]; then python3 $R/tools/rocpd_summary.py $DB 30 > $OUT/kernel_stats.txt; python3 $R/tools/rocpd_gaps.py $DB >> $OUT/kernel_stats.txt; fi
rm -rf $OUT/trace
cat $OUT/plain.json $OUT/traced.json
