#!/bin/bash
# Round-5 validation matrix of the precision rungs (tools/precision_ladder_sim.py --set round5): three geometries, five seeds,
# the reference's init scale and dense-block weights x 4.  CPU only (float64 emulation); ~2 h on 4 threads.
#   bash tools/ladder_round5.sh [outdir]      -> <outdir>/r5_<batch>x<size>_w<scale>.json (+ .log)
out=${1:-gpurun_out/sim}
mkdir -p "$out"
only="INFER,TRAIN: fwd,weight gradients only,backward-data + weight,exact16x3"
run() {  # batch size wscale seeds
  python tools/precision_ladder_sim.py --blocks 23 --batch $1 --size $2 --wscale $3 --seeds $4 --set round5 --threads 4 \
      --only "$only" --json "$out/r5_$1x$2_w$3.json" > "$out/r5_$1x$2_w$3.log" 2>&1
}
run 1 24 1 11,12,13,14,15
run 1 24 4 11,12,13,14,15
run 2 64 1 11,12,13,14,15
run 1 128 1 11,12,13,14,15
run 2 64 4 11,12
