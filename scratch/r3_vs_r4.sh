#!/bin/bash
# Round 3's final tree (git worktree add -f scratch/_r3 ee3ad87; build its library there) against HEAD, five interleaved rounds of
# bench.py on one box, batch 8 and batch 4.   scratch/r3_vs_r4.sh <outdir under gpurun_out>   (remove the worktree afterwards)
OUT=$PWD/gpurun_out/$1; mkdir -p $OUT
for i in 1 2 3 4 5; do
  for b in 8 4; do
    (cd scratch/_r3 && python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-kernel-profile > $OUT/r3_b${b}_$i.json 2> $OUT/r3_b${b}_$i.err)
    python bench.py --batch $b --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-kernel-profile --no-b4-leg > $OUT/r4_b${b}_$i.json 2> $OUT/r4_b${b}_$i.err
    python - <<PY
import json
for t in ("r3","r4"):
    j=[json.loads(l) for l in open("$OUT/%s_b${b}_$i.json" % t) if l.startswith("{")][0]
    print(t, "B=$b run=$i", j["value"], "img/s", j["ms_per_step"], "ms", flush=True)
PY
  done
done
