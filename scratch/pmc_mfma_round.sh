#!/bin/bash
# SQ / MFMA-utilisation counters over the six conv kernels of the 256-channel DoubleConv (scratch/kprof2.py), one --pmc pass.
#   scratch/pmc_mfma_round.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pmc -- python3 $R/scratch/kprof2.py > $OUT/pmc.log 2>&1 || { echo "pmc pass failed"; tail -5 $OUT/pmc.log; exit 1; }
echo "pmc pass done"; find $OUT -name "*.db" | head -3
