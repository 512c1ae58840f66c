#!/bin/bash
# final GPU call of round 5: the whole GPU suite, smoke(), the profile round on the final binary
set -o pipefail
mkdir -p gpurun_out/r5h
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r5h/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -n 6 gpurun_out/r5h/gpu_tests.log
python __graft_entry__.py smoke > gpurun_out/r5h/smoke.log 2>&1; echo "smoke rc=$?"; grep smoke gpurun_out/r5h/smoke.log
bash scratch/r5_profile_round.sh r5h/prof 2>&1 | tail -12
