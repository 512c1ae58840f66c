#!/bin/bash
# backward-weights + its slab reduction: the LDS-staged scale table against round 4's bf16 slabs (scratch/libs/libunet_hip_r4slabs.so), standalone and in the step
mkdir -p gpurun_out/r5j
R4=$PWD/scratch/libs/libunet_hip_r4slabs.so
python -m pytest tests/test_gpu_ops.py tests/test_gpu_wgrad_slabs.py tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "wgrad or slab or 16_bit or double_conv or batched" > gpurun_out/r5j/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 gpurun_out/r5j/tests.log
for i in 1 2; do
  python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5j/new_$i.txt; UH_LIB_PATH=$R4 python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5j/r4_$i.txt
  tail -n 1 gpurun_out/r5j/new_$i.txt; tail -n 1 gpurun_out/r5j/r4_$i.txt
done
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg > gpurun_out/r5j/new_$i.json 2> gpurun_out/r5j/new_$i.err
  UH_LIB_PATH=$R4 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg > gpurun_out/r5j/r4_$i.json 2> gpurun_out/r5j/r4_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5j/*.json")):
    j = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    print(f.split("/")[-1], j["value"], j["ms_per_step"], "b4", j["per_gpu_batch4"]["images_per_sec"], "wgrad family ms", j["kernels"]["conv3x3_wgrad_mfma"]["ms"])
PY
