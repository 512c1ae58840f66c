#!/bin/bash
# fp32 headline configuration: does the N = 1 global-batch-32 leg (2 GiB fp32 tensors) run?
mkdir -p gpurun_out/r5oc4
timeout -k 10 800 python bench.py --fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-inference > gpurun_out/r5oc4/fp32.json 2> gpurun_out/r5oc4/fp32.err; echo rc=$?
python - <<'PY'
import json
j = [json.loads(l) for l in open("gpurun_out/r5oc4/fp32.json") if l.startswith("{")][0]
print(j["value"], j["ms_per_step"], {k: v for k, v in (j.get("strong_gb32") or {}).items() if k != "double_conv_256_in_step"}, (j.get("per_gpu_batch4") or {}).get("images_per_sec"))
PY
tail -3 gpurun_out/r5oc4/fp32.err
