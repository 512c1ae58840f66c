#!/bin/bash
# like ab_env.sh but prints the stem / BatchNorm families too.   scratch/ab_env2.sh VAR outdir
VAR=$1; OUT=gpurun_out/$2; shift 2
mkdir -p $OUT
for i in 1 2; do for f in 0 1; do
  env $VAR=$f python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained "$@" > $OUT/ab_${f}_${i}.json 2> $OUT/ab_${f}_${i}.err
  python - <<PY
import json
j = json.load(open("$OUT/ab_${f}_${i}.json"))
k = j["kernels"]
print("$VAR=$f run=$i", j["value"], "img/s", j["ms_per_step"], "ms", {n: (v["calls"], v["ms"]) for n, v in k.items() if "calls" in v})
PY
done; done
