#!/bin/bash
# Kernel trace of the train step at the given per-GPU batches: per-kernel totals + one row per dispatch.
#   scratch/r4_trace.sh <outdir under gpurun_out> <batch> [<batch> ...]   (extra bench flags via $BENCH_EXTRA)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in "$@"; do
rocprofv3 --kernel-trace -d $OUT/stats_b$b -- python3 $R/bench.py --batch $b $BENCH_EXTRA --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained --no-b4-leg --steps 10 --warmup 3 > $OUT/stats_b$b.log 2>&1 || { echo "trace failed b$b"; tail -5 $OUT/stats_b$b.log; exit 1; }
db=$(find $OUT/stats_b$b -name "*.db" | head -1)
python3 $R/scratch/rocpd_export.py stats $db $OUT/kernel_stats_b$b.csv
python3 $R/scratch/rocpd_export.py trace $db $OUT/kernel_trace_b$b.csv
rm -rf $OUT/stats_b$b
echo done b$b
done
