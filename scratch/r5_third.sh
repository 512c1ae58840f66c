#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5c
python scratch/r5_dbg_slabs.py > gpurun_out/r5c/dbg_slabs.log 2>&1; echo "dbg rc=$?"; cat gpurun_out/r5c/dbg_slabs.log | grep -v amdgpu.ids
bash scratch/r5_ablate.sh r5c/abl 8 2>&1 | tail -45
