#!/bin/bash
# The other BASELINE configurations / batch sizes on the current binary (one gpurun call).   scratch/other_configs.sh <outdir>
OUT=gpurun_out/$1; mkdir -p $OUT
run() { name=$1; shift; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "$name FAILED: $(tail -2 $OUT/$name.err)";
  python - <<PY
import json
try:
    j = json.load(open("$OUT/$name.json"))
    s = j.get("sustained") or {}
    print("$name:", j["value"], "img/s", j["ms_per_step"], "ms/step; sustained", s.get("images_per_sec"), "| roofline", (j.get("roofline") or {}).get("frac"), "| dc256", ((j.get("kernels") or {}).get("double_conv_256") or {}).get("all_six"), j.get("strong_gb32"), j.get("collective"))
except Exception as e:
    print("$name: no result", e)
PY
}
run b8
run convt --convt
run cfg4_b2 --config4 --batch 2
run cfg5_fp32_b4 --fp32 --convt --batch 4 --cc-loss
run cfg5_bf16x3_b4 --fp32 --bf16x3 --convt --batch 4 --cc-loss
run b4 --batch 4
run b16 --batch 16
run b32 --batch 32
run gloo2 --gpus 2 --backend gloo --share-gpu
