#!/bin/bash
# N builds of libunet_hip.so (scratch/ab/lib_<name>.so, same C ABI) interleaved in one gpurun call:  scratch/ab_lib3.sh <outdir> <rounds> <name>...
OUT=gpurun_out/$1; mkdir -p $OUT; N=$2; shift 2
for i in $(seq 1 $N); do for v in "$@"; do
  UH_LIB_PATH=$PWD/scratch/ab/lib_$v.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained > $OUT/${v}_${i}.json 2> $OUT/${v}_${i}.err
  python - <<PY
import json
j = json.load(open("$OUT/${v}_${i}.json"))
k = j["kernels"]
print("$v run=$i", j["value"], "img/s", j["ms_per_step"], "ms", {n: v["ms"] for n, v in k.items() if "calls" in v}, k["double_conv_256"]["all_six"]["tflops"])
PY
done; done
