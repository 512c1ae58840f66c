#!/bin/bash
# Round 2's final tree (git worktree add -f scratch/_r2 bd63afb; build its library there) against HEAD, five interleaved rounds of
# bench.py on one box.   scratch/r2_vs_r3.sh <outdir under gpurun_out>     (remove the worktree afterwards: it is not to be committed)
OUT=gpurun_out/$1; mkdir -p $OUT
for i in 1 2 3 4 5; do
  (cd scratch/_r2 && python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference > ../../$OUT/r2_$i.json 2> ../../$OUT/r2_$i.err)
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained > $OUT/r3_$i.json 2> $OUT/r3_$i.err
  python - <<PY
import json
for t in ("r2","r3"):
    j=[json.loads(l) for l in open("$OUT/%s_$i.json" % t) if l.startswith("{")][0]
    print(t, "run=$i", j["value"], "img/s", j["ms_per_step"], "ms")
PY
done
