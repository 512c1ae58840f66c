#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5d
python scratch/r5_dbg_slabs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5d/dbg_slabs.log
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r5d/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -n 25 gpurun_out/r5d/gpu_tests.log
python scratch/r5_find_copies.py > gpurun_out/r5d/copies.log 2>&1; echo "copies rc=$?"
