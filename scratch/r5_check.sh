#!/bin/bash
# What the driver runs at round end, in one call: the GPU suite, smoke(), the default bench line.
set -o pipefail
mkdir -p gpurun_out/r5v
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r5v/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -n 3 gpurun_out/r5v/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5v/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5v/smoke.log
python bench.py > gpurun_out/r5v/bench.json 2> gpurun_out/r5v/bench.err; echo "bench rc=$?"; head -c 700 gpurun_out/r5v/bench.json
