#!/bin/bash
# HBM traffic of the headline step's chained launches as a walk (default) and as a pinned pipeline (RESR_CHAIN_PIPE): does the
# pipeline keep the block's planes inside the XCD's L2?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_pipe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in walk pipe; do
  if [ $mode = pipe ]; then export RESR_CHAIN_PIPE=${PIPE:-5,7,9,11}; else unset RESR_CHAIN_PIPE; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/${mode}_$C -o p -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-parity-mode --steps 2 --warmup 1 --no-probe > /dev/null 2> $OUT/${mode}_$C.err
  done
  F=$(find $OUT/${mode}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  W=$(find $OUT/${mode}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_traffic.py $F $W > $OUT/traffic_$mode.json
  rm -rf $OUT/${mode}_FETCH_SIZE $OUT/${mode}_WRITE_SIZE
done
python3 - <<PY
import json
for m in ("walk", "pipe"):
    d = json.load(open("$OUT/traffic_%s.json" % m))
    for k, v in d.items():
        if "chain" in k: print(m, k, v)
PY
