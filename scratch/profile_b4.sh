#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python3 $R/bench.py --batch 4 --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained --steps 10 --warmup 3 > $OUT/stats.log 2>&1 && echo done
