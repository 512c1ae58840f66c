#!/bin/bash
# Per-layer A/B of the BatchNorm-backward sums inside backward-data (uh_conv3x3_dgrad_bnsum): all eight fused (base), each one
# switched off alone, none fused -- interleaved, twice, one box.     scratch/r4_bnsum_layers.sh <outdir> [batch]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; B=${2:-8}
Q="--batch $B --no-cpu-baseline --no-inference --no-sustained --no-kernel-profile --no-b4-leg --steps 60 --warmup 10"
for i in 1 2; do
 for off in none 256x128 128x256 64x512 32x512 64x256 128x128 256x64 512x64 all; do
  if [ $off = all ]; then export UH_FUSE_BNSUM=0; unset UH_BNSUM_OFF; elif [ $off = none ]; then export UH_FUSE_BNSUM=1; unset UH_BNSUM_OFF; else export UH_FUSE_BNSUM=1; export UH_BNSUM_OFF=$off; fi
  python $R/bench.py $Q > $OUT/off_${off}_$i.json 2> $OUT/off_${off}_$i.err
  python - <<PY
import json
j=[json.loads(l) for l in open("$OUT/off_${off}_$i.json") if l.startswith("{")][0]
print("B=$B off=$off run=$i", j["value"], "img/s", j["ms_per_step"], "ms", flush=True)
PY
 done
done
