#!/bin/bash
# Round profile set on the GPU box (one gpurun call): kernel stats of the train step + the two HBM-traffic PMC passes.
#   scratch/profile_round.sh <outdir under gpurun_out>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained"
rocprofv3 --kernel-trace --stats -d $OUT/stats -- $BENCH --steps 10 --warmup 3 > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/pmc_write.log; exit 1; }
echo "write pass done"
find $OUT -name "*.csv" | head -20
du -sh $OUT
