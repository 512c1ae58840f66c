#!/bin/bash
# Round 4, first call: where does a batch-4 step (config 3's per-GPU workload) spend its time?  bench at B=8 / B=4, eager and as a
# captured graph, then a kernel trace at B=4 and B=8.   scratch/r4_first.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
Q="--no-cpu-baseline --no-inference --no-sustained --no-kernel-profile --steps 40 --warmup 8"
for i in 1 2; do
 for cfg in "8:" "4:" "8:--graph" "4:--graph" "2:" "2:--graph"; do
  b=${cfg%%:*}; x=${cfg#*:}
  python $R/bench.py --batch $b $x $Q > $OUT/b${b}${x}_$i.json 2> $OUT/b${b}${x}_$i.err
  python - <<PY
import json
j=[json.loads(l) for l in open("$OUT/b${b}${x}_$i.json") if l.startswith("{")][0]
print("B=$b $x run=$i", j["value"], "img/s", j["ms_per_step"], "ms", flush=True)
PY
 done
done
cd /tmp && export TMPDIR=/tmp
for b in 4 8; do
rocprofv3 --kernel-trace --stats -d $OUT/stats_b$b -- python3 $R/bench.py --batch $b --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained --steps 10 --warmup 3 > $OUT/stats_b$b.log 2>&1 && echo done b$b
f=$(ls $OUT/stats_b$b/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats_b$b.csv
rm -rf $OUT/stats_b$b
done
