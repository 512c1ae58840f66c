#!/bin/bash
# A/B of one environment switch inside ONE gpurun call, interleaved:   scratch/r4_ab_env.sh <outdir> VAR valA valB [rounds] [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; VAR=$2; A=$3; B=$4; N=${5:-3}; shift 5
cd $R
for i in $(seq 1 $N); do for v in $A $B; do
  env $VAR=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained "$@" > $OUT/${VAR}_${v}_${i}.json 2> $OUT/${VAR}_${v}_${i}.err
  python - <<PY
import json
j = [json.loads(l) for l in open("$OUT/${VAR}_${v}_${i}.json") if l.startswith("{")][0]
k = j.get("kernels") or {}
b4 = j.get("per_gpu_batch4") or {}
print("$VAR=$v run=$i", j["value"], "img/s", j["ms_per_step"], "ms", "| b4", b4.get("images_per_sec"), "|", {n: v["ms"] for n, v in k.items() if "calls" in v},
      "dc256", (k.get("double_conv_256") or {}).get("all_six", {}).get("tflops"), "in-step", (k.get("double_conv_256_in_step") or {}).get("all_six", {}).get("tflops"), flush=True)
PY
done; done
