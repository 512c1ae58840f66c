#!/bin/bash
# PMC passes over the 64->64 512^2 conv (no statistics = the backward-data launch): lib A = shipped kernel (a wave stores 32 contiguous
# bytes per pixel), lib B = the exchange epilogue (scratch/conv3x3_exch_epilogue.diff: every store instruction = 1 KiB of whole pixels).
#   scratch/pmc_store_exch.sh <outdir under gpurun_out>      (needs scratch/ab/lib_base.so, scratch/ab/lib_exch.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TA_BUSY_avr SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
P2="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCP_PENDING_STALL_CYCLES_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"
rocprofv3 --kernel-trace --pmc $P1 -d $OUT/p1 -- python3 $R/scratch/pmc_store_probe.py $R/scratch/ab/lib_base.so $R/scratch/ab/lib_exch.so > $OUT/p1.log 2>&1 || { echo "pass 1 failed"; tail -5 $OUT/p1.log; exit 1; }
echo "pass 1 done"
rocprofv3 --kernel-trace --pmc $P2 -d $OUT/p2 -- python3 $R/scratch/pmc_store_probe.py $R/scratch/ab/lib_base.so $R/scratch/ab/lib_exch.so > $OUT/p2.log 2>&1 || { echo "pass 2 failed"; tail -5 $OUT/p2.log; exit 1; }
echo "pass 2 done"
find $OUT -name "*.db" | head
