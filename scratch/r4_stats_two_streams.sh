#!/bin/bash
# Kernel stats of the train step exactly as the timed steps run it (backward-weights on the side stream), beside the one-stream stats of
# scratch/r4_profile_round.sh.   scratch/r4_stats_two_streams.sh <outdir>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/st -- python3 $R/bench.py --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained --no-b4-leg --steps 10 --warmup 3 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
db=$(find $OUT/st -name "*.db" | head -1)
python3 $R/scratch/rocpd_export.py stats $db $OUT/kernel_stats_b8_two_streams.csv; python3 $R/scratch/rocpd_export.py trace $db $OUT/kernel_trace_b8_two_streams.csv; rm -rf $OUT/st
python3 $R/scratch/r4_overlap.py $OUT/kernel_trace_b8_two_streams.csv 13 | head -3
head -8 $OUT/kernel_stats_b8_two_streams.csv | cut -c1-150
