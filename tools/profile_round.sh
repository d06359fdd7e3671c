#!/bin/bash
# Profiles of one round on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r03
# 1. rocprofv3 --kernel-trace --stats of the bench command (fast mode, then exact16) -> per-kernel tables
# 2. PMC passes in their own runs (FETCH_SIZE / WRITE_SIZE separately: they do not fit one pass; SQ stall counters)
# Everything lands under gpurun_out/prof_<tag>/; copy the reduced files you want judged into profiles/.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-parity-mode --no-sustained"
for P in fast exact16; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$P -o t -- $BENCH --precision $P --steps 5 --warmup 2 > $OUT/bench_trace_$P.json 2> $OUT/bench_trace_$P.err
  DB=$(find $OUT/trace_$P -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 $R/tools/rocpd_summary.py $DB 30 > $OUT/kernel_stats_$P.txt; python3 $R/tools/rocpd_gaps.py $DB >> $OUT/kernel_stats_$P.txt; rm -f $DB; fi
  find $OUT/trace_$P -name "*stats*.csv" -exec cp {} $OUT/kernel_stats_$P.csv \; 2>/dev/null
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${P}_$C -o p -- $BENCH --precision $P --steps 2 --warmup 1 --no-probe > /dev/null 2> $OUT/pmc_${P}_$C.err
  done
  F=$(find $OUT/pmc_${P}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  W=$(find $OUT/pmc_${P}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ] && [ -n "$W" ]; then python3 $R/tools/pmc_traffic.py $F $W > $OUT/pmc_traffic_$P.json; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
      --output-format csv -d $OUT/pmc_${P}_SQ -o p -- $BENCH --precision $P --steps 2 --warmup 1 --no-probe > /dev/null 2> $OUT/pmc_${P}_SQ.err
  S=$(find $OUT/pmc_${P}_SQ -name "*counter_collection.csv" | head -1)
  if [ -n "$S" ]; then python3 $R/tools/pmc_sq.py $S > $OUT/pmc_sq_$P.json; fi
  rm -rf $OUT/pmc_${P}_FETCH_SIZE $OUT/pmc_${P}_WRITE_SIZE $OUT/pmc_${P}_SQ $OUT/trace_$P
done
ls -la $OUT
