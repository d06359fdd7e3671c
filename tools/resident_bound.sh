#!/bin/bash
# Upper bound of an LDS-resident dense block: generator forward (train mode: six-job chained launches) with the
# in-tree library against the RESR_TIMING_RESIDENT timing build (tools/build_variant.py resident -DRESR_TIMING_RESIDENT=1), alternated.
for i in 1 2; do
  for geo in "32 64" "16 64"; do
    set -- $geo
    echo -n "in-tree   : "; python3 tools/fwd_loop.py --batch $1 --res $2 --train 2>/dev/null | tail -1
    echo -n "resident  : "; RESR_LIB_PATH=$PWD/tools/ab/resident.so python3 tools/fwd_loop.py --batch $1 --res $2 --train 2>/dev/null | tail -1
  done
done
