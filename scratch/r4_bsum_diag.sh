#!/bin/bash
# Where the backward-data + BatchNorm-sums launches spend their extra time: the default library against two diagnostic variants built
# on the box (UH_X_BSUM=1: no q traffic; =2: q fetched, sums not formed -- both give WRONG sums, timing only), interleaved.
#   scratch/r4_bsum_diag.sh <outdir> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; N=${2:-2}
cd $R
for x in 1 2; do python scratch/mkvariant.py xb$x -DUH_X_BSUM=$x > $OUT/build_xb$x.log 2>&1 || { tail -20 $OUT/build_xb$x.log; exit 1; }; done
for i in $(seq 1 $N); do for v in base xb1 xb2 nofuse; do
  unset UH_LIB_PATH UH_FUSE_BNSUM
  case $v in xb1|xb2) export UH_LIB_PATH=$R/scratch/variants/libunet_hip_$v.so;; nofuse) export UH_FUSE_BNSUM=0;; esac
  python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-inference --no-sustained --no-b4-leg > $OUT/${v}_${i}.json 2> $OUT/${v}_${i}.err
  python - <<PY
import json
j = [json.loads(l) for l in open("$OUT/${v}_${i}.json") if l.startswith("{")][0]
k = j.get("kernels") or {}
print("$v run=$i", j["value"], "img/s", j["ms_per_step"], "ms |", {n: v["ms"] for n, v in k.items() if "calls" in v}, flush=True)
PY
done; done
