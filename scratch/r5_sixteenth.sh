#!/bin/bash
# Cache policy of the conv output stores: default against nt (aux 2), sc0 (aux 1), sc1 (aux 16) -- scratch/r5_mklib.py builds of the same
# source with -DUH_ST_AUX=n.  Conv tests on the nt build first, then every conv launch shape, then the step.
mkdir -p gpurun_out/r5y
P=$PWD/scratch/libs
UH_LIB_PATH=$P/libunet_hip_staux2.so timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv3x3" > gpurun_out/r5y/tests.log 2>&1
rc=$?; echo "nt tests rc=$rc"; tail -2 gpurun_out/r5y/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
  python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5y/base_$i.txt || exit 1
  for a in 2 1 16; do
    UH_LIB_PATH=$P/libunet_hip_staux$a.so python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5y/aux${a}_$i.txt || exit 1
  done
done
for i in 1 2; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg --no-b4-leg 2>gpurun_out/r5y/bb_$i.err > gpurun_out/r5y/bb_$i.json &&
  UH_LIB_PATH=$P/libunet_hip_staux2.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg --no-b4-leg 2>gpurun_out/r5y/bn_$i.err > gpurun_out/r5y/bn_$i.json || exit 1
done
python - <<'PY'
import re, json
order = ["base_1", "aux2_1", "aux1_1", "aux16_1", "base_2", "aux2_2", "aux1_2", "aux16_2"]
rows = {}
for v in order:
    for line in open("gpurun_out/r5y/%s.txt" % v):
        m = re.match(r"(\S+)\s+H=\s*(\d+)\s+(\d+) ->\s*(\d+) \| fwd\s+([\d.]+) us .*\| dgrad\s+([\d.]+) us", line)
        if m: rows.setdefault(m.group(1), {})[v] = (float(m.group(5)), float(m.group(6)))
print("us per launch; columns:", order)
for k, d in rows.items():
    print(f"{k:8s} fwd  ", " ".join(f"{d[v][0]:7.1f}" if v in d else "      -" for v in order))
    print(f"{k:8s} dgrad", " ".join(f"{d[v][1]:7.1f}" if v in d else "      -" for v in order))
print("sum     ", " ".join(f"{sum(d[v][0] + d[v][1] for d in rows.values() if v in d):7.0f}" for v in order))
for i in (1, 2):
    for t in ("bb", "bn"):
        j = [json.loads(l) for l in open(f"gpurun_out/r5y/{t}_{i}.json") if l.startswith("{")][0]
        print(t, i, j["value"], "img/s")
PY
