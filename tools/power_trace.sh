#!/bin/bash
# GPU power / clock samples (rocm-smi) while the headline bench runs: is the step at the board's power cap?
OUT=${1:-gpurun_out/power_trace.txt}
rocm-smi --showmaxpower --showpower --showclocks > $OUT 2>&1
python bench.py --no-cpu-baseline --no-parity-mode --no-other-configs --steps 200 --warmup 5 > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 40); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|memory)" | tr '\n' ' ' >> $OUT; echo >> $OUT
  sleep 0.5
done
wait $BP
tail -1 gpurun_out/power_bench.json | cut -c1-160 >> $OUT
cat $OUT
