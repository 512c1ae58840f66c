#!/bin/bash
# XCD-contiguous tile lanes (-DUH_XCD_BLOCK=1, scratch/libs/libunet_hip_xcdblock.so) against the round-robin mapping
mkdir -p gpurun_out/r5k
X=$PWD/scratch/libs/libunet_hip_xcdblock.so
for i in 1 2; do
  python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5k/base_$i.txt; UH_LIB_PATH=$X python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5k/xcd_$i.txt
  tail -n 1 gpurun_out/r5k/base_$i.txt; tail -n 1 gpurun_out/r5k/xcd_$i.txt
done
paste <(cut -c1-70 gpurun_out/r5k/base_1.txt) <(cut -c26-70 gpurun_out/r5k/xcd_1.txt) | head -20
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg > gpurun_out/r5k/base_$i.json 2> gpurun_out/r5k/base_$i.err
  UH_LIB_PATH=$X python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg > gpurun_out/r5k/xcd_$i.json 2> gpurun_out/r5k/xcd_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5k/*.json")):
    j = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    print(f.split("/")[-1], j["value"], j["ms_per_step"], "b4", j["per_gpu_batch4"]["images_per_sec"], "fwd family ms", j["kernels"]["conv3x3_fwd_mfma"]["ms"])
PY
