#!/bin/bash
# The PMC passes of tools/profile_round.sh alone (FETCH_SIZE / WRITE_SIZE / SQ counters, each in its own run, no tracing):  bash tools/pmc_round.sh r06b [fast exact16]
set -u
TAG=${1:-r06b}; shift
PRECS=${*:-fast exact16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-parity-mode --no-sustained"
for P in $PRECS; do
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
  rm -rf $OUT/pmc_${P}_FETCH_SIZE $OUT/pmc_${P}_WRITE_SIZE $OUT/pmc_${P}_SQ
done
ls -la $OUT
