#!/bin/bash
# stored stem / serial-FMA recompute / MFMA recompute, interleaved twice in one gpurun call
OUT=gpurun_out/$1; mkdir -p $OUT
for i in 1 2; do for v in stored valu mfma; do
  case $v in stored) E="UH_STEM_RECOMPUTE=0";; valu) E="UH_STEM_VALU=1";; mfma) E="UH_X=0";; esac
  env $E python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained > $OUT/${v}_${i}.json 2> $OUT/${v}_${i}.err
  python - <<PY
import json
j = json.load(open("$OUT/${v}_${i}.json"))
k = j["kernels"]
print("$v run=$i", j["value"], "img/s", j["ms_per_step"], "ms", {n: (v["calls"], v["ms"]) for n, v in k.items() if "calls" in v and "stem" in n})
PY
done; done
