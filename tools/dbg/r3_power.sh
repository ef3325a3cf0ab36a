#!/bin/bash
# power / clock under load: rocm-smi sampled while (a) the fused multiplier pass runs at 1, 2 and 4 waves per SIMD, (b) the bench runs
mkdir -p gpurun_out/r3
O=gpurun_out/r3/power.txt
: > $O
sample() { for k in $(seq 1 $1); do echo "--- $2 t=$k" >> $O; rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" >> $O; sleep 0.25; done; }
rocm-smi --showpower --showclocks >> $O 2>&1
for w in 1 2 4; do
  ./tools/ubench/ubench3 long $w >> gpurun_out/r3/power_ubench.txt 2>&1 &
  pid=$!
  sleep 0.6
  sample 6 "ubench wps=$w"
  wait $pid
done
python bench.py --steps 200 --warmup 3 --no-extras > gpurun_out/r3/bench_long.json 2> gpurun_out/r3/bench_long.err &
pid=$!
sleep 6
sample 10 "bench"
wait $pid
tail -c 600 gpurun_out/r3/bench_long.json | head -c 300
