#!/bin/bash
# Cache policy of the halo LDS-DMA (forward / backward-data AND backward-weights share the helper): default against nt and sc1.
mkdir -p gpurun_out/r5z
P=$PWD/scratch/libs
UH_LIB_PATH=$P/libunet_hip_dmant.so timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv3x3" > gpurun_out/r5z/tests.log 2>&1
rc=$?; echo "nt tests rc=$rc"; tail -2 gpurun_out/r5z/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
  python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5z/base_$i.txt || exit 1
  for a in dmant dmasc1; do
    UH_LIB_PATH=$P/libunet_hip_$a.so python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5z/${a}_$i.txt || exit 1
  done
done
for i in 1 2; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg --no-b4-leg 2>gpurun_out/r5z/bb_$i.err > gpurun_out/r5z/bb_$i.json &&
  UH_LIB_PATH=$P/libunet_hip_dmant.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg --no-b4-leg 2>gpurun_out/r5z/bn_$i.err > gpurun_out/r5z/bn_$i.json || exit 1
done
python - <<'PY'
import re, json
order = ["base_1", "dmant_1", "dmasc1_1", "base_2", "dmant_2", "dmasc1_2"]
rows = {}
for v in order:
    for line in open("gpurun_out/r5z/%s.txt" % v):
        m = re.match(r"(\S+)\s+H=\s*(\d+)\s+(\d+) ->\s*(\d+) \| fwd\s+([\d.]+) us .*\| dgrad\s+([\d.]+) us .*\| wgrad\+reduce\s+([\d.]+) us", line)
        if m: rows.setdefault(m.group(1), {})[v] = (float(m.group(5)), float(m.group(6)), float(m.group(7)))
print("us per launch; columns:", order)
for k, d in rows.items():
    for pi, pn in enumerate(("fwd", "dgrad", "wgrad")):
        print(f"{k:8s} {pn:5s}", " ".join(f"{d[v][pi]:7.1f}" if v in d else "      -" for v in order))
print("sum     ", " ".join(f"{sum(sum(d[v]) for d in rows.values() if v in d):7.0f}" for v in order))
for i in (1, 2):
    for t in ("bb", "bn"):
        j = [json.loads(l) for l in open(f"gpurun_out/r5z/{t}_{i}.json") if l.startswith("{")][0]
        print(t, i, j["value"], "img/s")
PY
