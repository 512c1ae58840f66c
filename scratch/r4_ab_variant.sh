#!/bin/bash
# The default library against a variant built ON THE BOX with extra -D flags (scratch/variants/ does not travel), interleaved:
#   scratch/r4_ab_variant.sh <outdir> <name> "<-D flags>" [rounds] [extra bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; NAME=$2; FLAGS=$3; N=${4:-3}; shift 4
cd $R
python scratch/mkvariant.py $NAME $FLAGS > $OUT/build_$NAME.log 2>&1 || { tail -20 $OUT/build_$NAME.log; exit 1; }
VAR=$R/scratch/variants/libunet_hip_$NAME.so
for i in $(seq 1 $N); do for v in base $NAME; do
  if [ $v = base ]; then unset UH_LIB_PATH; else export UH_LIB_PATH=$VAR; fi
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained "$@" > $OUT/${v}_${i}.json 2> $OUT/${v}_${i}.err
  python - <<PY
import json
j = [json.loads(l) for l in open("$OUT/${v}_${i}.json") if l.startswith("{")][0]
k = j.get("kernels") or {}
b4 = j.get("per_gpu_batch4") or {}
print("$v run=$i", j["value"], "img/s", j["ms_per_step"], "ms", "| b4", b4.get("images_per_sec"), "|", {n: v["ms"] for n, v in k.items() if "calls" in v},
      "dc256", (k.get("double_conv_256") or {}).get("all_six", {}).get("tflops"), "in-step", (k.get("double_conv_256_in_step") or {}).get("all_six", {}).get("tflops"), flush=True)
PY
done; done
