#!/bin/bash
# halo LDS-DMA with sc1 against the default policy on the step (four interleaved pairs) -- the launch shapes gave -0.9 / -1.3 %
mkdir -p gpurun_out/r5z2
P=$PWD/scratch/libs
for i in 1 2 3 4; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg 2>gpurun_out/r5z2/bb_$i.err > gpurun_out/r5z2/bb_$i.json &&
  UH_LIB_PATH=$P/libunet_hip_dmasc1.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg 2>gpurun_out/r5z2/bn_$i.err > gpurun_out/r5z2/bn_$i.json || exit 1
done
python - <<'PY'
import json
for i in (1, 2, 3, 4):
    for t in ("bb", "bn"):
        j = [json.loads(l) for l in open(f"gpurun_out/r5z2/{t}_{i}.json") if l.startswith("{")][0]
        print(t, i, j["value"], "img/s | b4", j["per_gpu_batch4"]["images_per_sec"], "| conv fwd frac", j["roofline"]["frac"])
PY
