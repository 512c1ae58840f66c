#!/bin/bash
# round 5, second GPU call: the whole GPU suite on the new binary, A/B against round 4's bf16 slabs, the WRES-wide dispatch experiment
set -o pipefail
mkdir -p gpurun_out/r5b
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/r5b/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -n 5 gpurun_out/r5b/gpu_tests.log
python scratch/r5_find_copies.py > gpurun_out/r5b/copies.log 2>&1; echo "copies rc=$?"
R4=$PWD/scratch/libs/libunet_hip_r4slabs.so
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained > gpurun_out/r5b/new_$i.json 2> gpurun_out/r5b/new_$i.err; echo "new $i rc=$?"
  UH_LIB_PATH=$R4 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained > gpurun_out/r5b/r4slabs_$i.json 2> gpurun_out/r5b/r4slabs_$i.err; echo "r4 $i rc=$?"
  UH_WRES_WIDE=1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained > gpurun_out/r5b/wreswide_$i.json 2> gpurun_out/r5b/wreswide_$i.err; echo "wide $i rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5b/*.json")):
    try:
        j = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    except Exception as e:
        print(f, "unreadable", e); continue
    k = j.get("kernels") or {}
    b4 = j.get("per_gpu_batch4") or {}
    print(f.split("/")[-1], j["value"], "img/s", j["ms_per_step"], "ms | b4", b4.get("images_per_sec"), "| frac", j["roofline"]["frac"],
          {n: v["ms"] for n, v in k.items() if isinstance(v, dict) and "calls" in v})
PY
