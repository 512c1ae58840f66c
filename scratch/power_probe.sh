#!/bin/bash
# samples board power / clocks of every GPU rocm-smi sees while bench.py runs; prints the samples above 500 W
python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-inference > gpurun_out/pp_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/\t//g' | paste - - | awk -F'[:()]' '{ if ($NF+0 > 500 || $(NF-0)+0 > 500) print }' 
  sleep 0.5
done
tail -1 gpurun_out/pp_bench.log | cut -c1-200
