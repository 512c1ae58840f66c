#!/bin/bash
# Kernel trace of the train step with backward-weights on the side stream: which launches overlap, and for how long?   scratch/r4_side_trace.sh <outdir>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/st -- python3 $R/bench.py --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained --no-b4-leg --side-stream --steps 6 --warmup 3 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
db=$(find $OUT/st -name "*.db" | head -1)
python3 $R/scratch/rocpd_export.py trace $db $OUT/kernel_trace_side.csv; rm -rf $OUT/st
python3 $R/scratch/r4_overlap.py $OUT/kernel_trace_side.csv 9
