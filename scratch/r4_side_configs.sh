#!/bin/bash
# The side stream for backward-weights on the OTHER configurations, with / without, interleaved (one gpurun call).  scratch/r4_side_configs.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
run() { name=$1; shift
  for rep in 1 2; do for v in one side; do
    F="--no-side-stream"; [ $v = side ] && F="--side-stream"
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-sustained --no-kernel-profile --no-b4-leg $F "$@" > $OUT/${name}_${v}_$rep.json 2> $OUT/${name}_${v}_$rep.err || echo "$name $v FAILED: $(tail -2 $OUT/${name}_${v}_$rep.err)"
    python - <<PY
import json
try:
    j = [json.loads(l) for l in open("$OUT/${name}_${v}_$rep.json") if l.startswith("{")][0]
    print("$name $v $rep:", j["value"], "img/s", j["ms_per_step"], "ms", flush=True)
except Exception as e:
    print("$name $v: no result", e)
PY
  done; done
}
run convt --convt
run cfg4_b2 --config4 --batch 2
run cfg5_fp32_b4 --fp32 --convt --batch 4 --cc-loss
run cfg5_bf16x3_b4 --fp32 --bf16x3 --convt --batch 4 --cc-loss
run b2 --batch 2
run b16 --batch 16
run b32 --batch 32
run gloo2 --gpus 2 --backend gloo --share-gpu
