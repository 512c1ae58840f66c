#!/bin/bash
# Would two tiles per workgroup at 16 channels per wave pay?  Probe without building it: the 16-channel-per-wave form forced on every layer
# (UH_FORCE_NBW1, probe library) with and without its filter-fragment loads skipped on every other tile (-DUH_ABL_HALFW=1: timing only,
# results garbage) against the shipped selection.  scratch/r5/conv_tw2_probe.diff is the source change the probe libraries carry.
mkdir -p gpurun_out/r5p
P=$PWD/scratch/libs
for i in 1 2; do
  python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5p/base_$i.txt
  UH_FORCE_NBW1=1 UH_LIB_PATH=$P/libunet_hip_probe_nbw1.so python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5p/nbw1_$i.txt
  UH_FORCE_NBW1=1 UH_LIB_PATH=$P/libunet_hip_probe_nbw1_halfw.so python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5p/nbw1_halfw_$i.txt
  UH_LIB_PATH=$P/libunet_hip_probe_nbw1_halfw.so python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5p/halfw_$i.txt
done
python - <<'PY'
import re
order = ["base_1", "base_2", "nbw1_1", "nbw1_2", "nbw1_halfw_1", "nbw1_halfw_2", "halfw_1", "halfw_2"]
rows = {}
for v in order:
    for line in open("gpurun_out/r5p/%s.txt" % v):
        m = re.match(r"(\S+)\s+H=\s*(\d+)\s+(\d+) ->\s*(\d+) \| fwd\s+([\d.]+) us .*\| dgrad\s+([\d.]+) us", line)
        if m: rows.setdefault(m.group(1), {})[v] = (float(m.group(5)), float(m.group(6)))
print("us per launch; columns:", order)
for k, d in rows.items():
    print(f"{k:8s} fwd  ", " ".join(f"{d[v][0]:7.1f}" if v in d else "      -" for v in order))
    print(f"{k:8s} dgrad", " ".join(f"{d[v][1]:7.1f}" if v in d else "      -" for v in order))
print("sum     ", " ".join(f"{sum(d[v][0] + d[v][1] for d in rows.values() if v in d):7.0f}" for v in order))
PY
