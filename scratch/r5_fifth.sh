#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5e
python -m pytest tests/test_gpu_dp.py tests/test_gpu_parity.py tests/test_gpu_ops.py tests/test_gpu_wgrad_slabs.py tests/test_gpu_bf16_vs_reference.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r5e/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -n 12 gpurun_out/r5e/gpu_tests.log
python bench.py > gpurun_out/r5e/bench.json 2> gpurun_out/r5e/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = [json.loads(l) for l in open("gpurun_out/r5e/bench.json") if l.startswith("{")][0]
print(j["value"], j["ms_per_step"], "b4", j.get("per_gpu_batch4", {}).get("images_per_sec"), j.get("per_gpu_batch4", {}).get("vs_batch8_images_per_sec"), "strong", j.get("strong_gb32"))
print("roofline", j["roofline"])
print("cpu", {k: v for k, v in j["cpu_baseline"].items() if k != "dice_vs_ref"})
print("dice", j["cpu_baseline"].get("dice_vs_ref"))
print("kernels", {n: v.get("ms") for n, v in j["kernels"].items() if isinstance(v, dict) and "calls" in v})
PY
