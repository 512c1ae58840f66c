#!/bin/bash
# What do the chunk-fence barriers cost?  Timing-only build without the s_barrier of the chunk fences (waits kept; results garbage by design:
# waves read halo pieces that other waves' DMA has not landed yet) against the shipped library, every conv launch shape of config 2.
mkdir -p gpurun_out/r5x
P=$PWD/scratch/libs
for i in 1 2; do
  python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5x/base_$i.txt &&
  UH_LIB_PATH=$P/libunet_hip_abl_NOBAR.so python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5x/nobar_$i.txt || exit 1
done
python - <<'PY'
import re
order = ["base_1", "nobar_1", "base_2", "nobar_2"]
rows = {}
for v in order:
    for line in open("gpurun_out/r5x/%s.txt" % v):
        m = re.match(r"(\S+)\s+H=\s*(\d+)\s+(\d+) ->\s*(\d+) \| fwd\s+([\d.]+) us .*\| dgrad\s+([\d.]+) us", line)
        if m: rows.setdefault(m.group(1), {})[v] = (float(m.group(5)), float(m.group(6)))
print("us per launch; columns:", order)
for k, d in rows.items():
    print(f"{k:8s} fwd  ", " ".join(f"{d[v][0]:7.1f}" if v in d else "      -" for v in order))
    print(f"{k:8s} dgrad", " ".join(f"{d[v][1]:7.1f}" if v in d else "      -" for v in order))
print("sum     ", " ".join(f"{sum(d[v][0] + d[v][1] for d in rows.values() if v in d):7.0f}" for v in order))
PY
