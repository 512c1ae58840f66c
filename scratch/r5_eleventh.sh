#!/bin/bash
# Early filter request (UH_W_EARLY): each of the three filter sets of the streaming conv forms is requested again right behind its
# last use, so it has two column shifts to land under instead of one.  Correctness first (conv tests under the variant library),
# then per-layer times and the step, interleaved with the shipped library.
mkdir -p gpurun_out/r5q
P=$PWD/scratch/libs
UH_LIB_PATH=$P/libunet_hip_wearly.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_bnsum.py tests/test_gpu_bf16_vs_reference.py -q -m gpu -x > gpurun_out/r5q/tests.log 2>&1
rc=$?; echo "variant tests rc=$rc"; tail -4 gpurun_out/r5q/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
  python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5q/base_$i.txt &&
  UH_LIB_PATH=$P/libunet_hip_wearly.so python scratch/r4_conv_bench.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r5q/wearly_$i.txt || exit 1
done
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg 2>gpurun_out/r5q/bb_$i.err > gpurun_out/r5q/bb_$i.json &&
  UH_LIB_PATH=$P/libunet_hip_wearly.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg 2>gpurun_out/r5q/bw_$i.err > gpurun_out/r5q/bw_$i.json || exit 1
done
python - <<'PY'
import re, json
order = ["base_1", "wearly_1", "base_2", "wearly_2"]
rows = {}
for v in order:
    for line in open("gpurun_out/r5q/%s.txt" % v):
        m = re.match(r"(\S+)\s+H=\s*(\d+)\s+(\d+) ->\s*(\d+) \| fwd\s+([\d.]+) us .*\| dgrad\s+([\d.]+) us", line)
        if m: rows.setdefault(m.group(1), {})[v] = (float(m.group(5)), float(m.group(6)))
print("us per launch; columns:", order)
for k, d in rows.items():
    print(f"{k:8s} fwd  ", " ".join(f"{d[v][0]:7.1f}" if v in d else "      -" for v in order))
    print(f"{k:8s} dgrad", " ".join(f"{d[v][1]:7.1f}" if v in d else "      -" for v in order))
print("sum     ", " ".join(f"{sum(d[v][0] + d[v][1] for d in rows.values() if v in d):7.0f}" for v in order))
for i in (1, 2, 3):
    for t in ("bb", "bw"):
        try:
            j = json.loads(open(f"gpurun_out/r5q/{t}_{i}.json").read().strip().splitlines()[-1])
            print(t, i, j["value"], "img/s", j["ms_per_step"], "ms | b4", j.get("b4", {}).get("value") if isinstance(j.get("b4"), dict) else None)
        except Exception as e: print(t, i, "unreadable", e)
PY
