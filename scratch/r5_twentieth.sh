#!/bin/bash
mkdir -p gpurun_out/r5oc3
timeout -k 10 600 python -m pytest tests/test_gpu_large.py -q -m gpu -x > gpurun_out/r5oc3/large.log 2>&1; echo "large rc=$?"; tail -3 gpurun_out/r5oc3/large.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --convt > gpurun_out/r5oc3/convt.json 2> gpurun_out/r5oc3/convt.err; echo "convt rc=$?"
python - <<'PY'
import json
j = [json.loads(l) for l in open("gpurun_out/r5oc3/convt.json") if l.startswith("{")][0]
print("convt:", j["value"], "img/s", j["ms_per_step"], "ms | sustained", (j.get("sustained") or {}).get("images_per_sec"), "| strong", {k: v for k, v in (j.get("strong_gb32") or {}).items() if k != "double_conv_256_in_step"}, "| b4", (j.get("per_gpu_batch4") or {}).get("images_per_sec"))
PY
