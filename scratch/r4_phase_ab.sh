#!/bin/bash
# Start-phase experiment on the forward / backward-data conv kernel (UH_FWD_STAGGER, UH_FWD_PRIO), interleaved in ONE gpurun call:
#   scratch/r4_phase_ab.sh <outdir> [rounds] -- configurations are "stagger:prio" pairs
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; N=${2:-2}
CFGS=${CFGS:-"0:0 4:0 12:0 24:0 0:1 12:1"}
cd $R
for i in $(seq 1 $N); do for c in $CFGS; do
  s=${c%%:*}; p=${c##*:}
  UH_FWD_STAGGER=$s UH_FWD_PRIO=$p python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-inference --no-sustained --no-b4-leg > $OUT/ph_${s}_${p}_${i}.json 2> $OUT/ph_${s}_${p}_${i}.err
  python - <<PY
import json
j = [json.loads(l) for l in open("$OUT/ph_${s}_${p}_${i}.json") if l.startswith("{")][0]
k = j.get("kernels") or {}
print("stagger=$s prio=$p run=$i", j["value"], "img/s", j["ms_per_step"], "ms |", {n: v["ms"] for n, v in k.items() if "calls" in v},
      "in-step", (k.get("double_conv_256_in_step") or {}).get("all_six", {}).get("tflops"), flush=True)
PY
done; done
