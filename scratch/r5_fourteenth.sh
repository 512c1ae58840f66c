#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5u
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r5u/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -n 8 gpurun_out/r5u/gpu_tests.log
bash scratch/r5_profile_round.sh r5u/prof 2>&1 | tail -15
