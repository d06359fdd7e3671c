#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the cout-32 dense-block passes at 16 x 64^2 -- six planes
# of two images per XCD are 3.1 MB, inside the 4 MB L2 -- as chained launches and as one launch per pass: where the planes fit,
# the chain's jobs re-read them from L2 (no kernel boundary in between), i.e. the bytes of the fusion the full-size batch cannot have.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_chain_small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in chain nochain; do
  if [ $mode = nochain ]; then export RESR_CONV_NO_CHAIN=1; else unset RESR_CONV_NO_CHAIN; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    B=${BATCH:-16} STEPS=2 rocprofv3 --pmc $C --output-format csv -d $OUT/${mode}_$C -o p -- python3 $R/tools/bench_cfg3.py > /dev/null 2> $OUT/${mode}_$C.err
  done
  F=$(find $OUT/${mode}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  W=$(find $OUT/${mode}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_traffic.py $F $W > $OUT/traffic_$mode.json
  rm -rf $OUT/${mode}_FETCH_SIZE $OUT/${mode}_WRITE_SIZE
done
python3 - <<PY
import json
for m in ("chain", "nochain"):
    d = json.load(open("$OUT/traffic_%s.json" % m))
    for k, v in d.items():
        if "conv3x3_ws_kernel<f16,1," in k:
            print(m, k, v)
PY
