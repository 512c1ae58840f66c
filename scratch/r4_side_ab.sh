#!/bin/bash
# backward-weights on a side stream (bench.py --side-stream) against the one-stream default, interleaved in one call:  scratch/r4_side_ab.sh <outdir> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; N=${2:-3}; cd $R
for i in $(seq 1 $N); do for v in one side; do
  F=""; [ $v = side ] && F="--side-stream"
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-kernel-profile $F > $OUT/${v}_$i.json 2> $OUT/${v}_$i.err
  python - <<PY
import json
j = [json.loads(l) for l in open("$OUT/${v}_$i.json") if l.startswith("{")][0]
b4 = j.get("per_gpu_batch4") or {}
print("$v run=$i", j["value"], "img/s", j["ms_per_step"], "ms | b4", b4.get("images_per_sec"), flush=True)
PY
done; done
