#!/bin/bash
# Round-5 profile set on the GPU box (one gpurun call): kernel trace of the train step at B=8 and B=4, the two HBM-traffic PMC
# passes, the SQ / MFMA-utilisation pass over the 256-channel DoubleConv, and the default bench line.  Everything is exported to
# csv on the box (rocprofv3 writes a rocpd database); the databases are not kept.    scratch/r5_profile_round.sh <outdir>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained --no-b4-leg --no-strong-leg --no-side-stream"      # (one stream: every kernel alone, durations comparable with the live events of bench.py)
exp() { db=$(find $1 -name "*.db" | head -1); python3 $R/scratch/rocpd_export.py $2 $db $3; rm -rf $1; }
for b in 8 4; do
  rocprofv3 --kernel-trace --stats -d $OUT/st$b -- $BENCH --batch $b --steps 10 --warmup 3 > $OUT/stats_b$b.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats_b$b.log; exit 1; }
  db=$(find $OUT/st$b -name "*.db" | head -1)
  python3 $R/scratch/rocpd_export.py stats $db $OUT/kernel_stats_b$b.csv; python3 $R/scratch/rocpd_export.py trace $db $OUT/kernel_trace_b$b.csv; rm -rf $OUT/st$b
  echo "stats pass b$b done"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pf -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
mkdir -p $OUT/pmc_fetch; exp $OUT/pf counters $OUT/pmc_fetch; echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pw -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/pmc_write.log; exit 1; }
mkdir -p $OUT/pmc_write; exp $OUT/pw counters $OUT/pmc_write; echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pm -- python3 $R/scratch/kprof2.py > $OUT/pmc_mfma.log 2>&1 || { echo "mfma pass failed"; tail -5 $OUT/pmc_mfma.log; exit 1; }
mkdir -p $OUT/pmc_mfma; exp $OUT/pm counters $OUT/pmc_mfma; echo "mfma pass done"
cd $R && python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err; tail -c 600 $OUT/bench_line.json
du -sh $OUT
