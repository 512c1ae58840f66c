#!/bin/bash
# Turn a scratch/r4_profile_round.sh output directory (gpurun_out/<dir>) into the committed profiles/r04_* artefacts.
#   scratch/r4_make_profiles.sh <dir under gpurun_out>
D=gpurun_out/$1
cp $D/kernel_stats_b8.csv profiles/r04_a_bench_kernel_stats_b8.csv; cp $D/kernel_stats_b4.csv profiles/r04_a_bench_kernel_stats_b4.csv; cp $D/bench_line.json profiles/r04_a_bench_line.json
V=$(python -c "import json;print([json.loads(l) for l in open('$D/bench_line.json') if l.startswith('{')][0]['value'])")
python scratch/conv_layer_table.py $D/kernel_trace_b8.csv 13 profiles/r04_conv_layer_table.md "bench.py on the same box: $V images/s (profiles/r04_a_bench_line.json)" | tail -1
python scratch/pmc_traffic2.py $D/pmc_fetch $D/pmc_write 3 profiles/r04_hbm_traffic_pmc.json > /tmp/traffic.log 2>&1
python scratch/pmc_mfma_util.py $D/pmc_mfma > profiles/r04_pmc_sq_mfma_util_double_conv_256.csv
