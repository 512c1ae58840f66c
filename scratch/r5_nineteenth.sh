#!/bin/bash
# bench.py --convt failed at N = 1 since the global-batch-32 leg runs there too (32 x 512 x 512 passes the transposed-conv kernels' 2 GiB limit):
# the leg now records the refusal and the line is printed.  Also the gloo 2-rank line once more.
mkdir -p gpurun_out/r5oc2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --convt > gpurun_out/r5oc2/convt.json 2> gpurun_out/r5oc2/convt.err; echo "convt rc=$?"
python - <<'PY'
import json
j = [json.loads(l) for l in open("gpurun_out/r5oc2/convt.json") if l.startswith("{")][0]
print("convt:", j["value"], "img/s", j["ms_per_step"], "ms | sustained", (j.get("sustained") or {}).get("images_per_sec"), "| strong", j.get("strong_gb32"), "| b4", (j.get("per_gpu_batch4") or {}).get("images_per_sec"))
PY
