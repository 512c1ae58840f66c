#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5m
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r5m/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -n 4 gpurun_out/r5m/gpu_tests.log
bash scratch/r5_profile_round.sh r5m/prof 2>&1 | tail -8 | cut -c1-300
