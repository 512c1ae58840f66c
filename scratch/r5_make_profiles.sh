#!/bin/bash
# Turn a scratch/r5_profile_round.sh output directory (gpurun_out/<dir>) into the committed profiles/r05_* artefacts.
#   scratch/r5_make_profiles.sh <dir under gpurun_out> [tag, default a]
D=gpurun_out/$1; T=${2:-a}
cp $D/kernel_stats_b8.csv profiles/r05_${T}_bench_kernel_stats_b8.csv; cp $D/kernel_stats_b4.csv profiles/r05_${T}_bench_kernel_stats_b4.csv; cp $D/bench_line.json profiles/r05_${T}_bench_line.json
V=$(python -c "import json;print([json.loads(l) for l in open('$D/bench_line.json') if l.startswith('{')][0]['value'])")
python scratch/conv_layer_table.py $D/kernel_trace_b8.csv 13 profiles/r05_${T}_conv_layer_table.md "bench.py on the same box: $V images/s (profiles/r05_${T}_bench_line.json)" | tail -1
python scratch/conv_layer_table.py $D/kernel_trace_b4.csv 13 profiles/r05_${T}_conv_layer_table_b4.md "batch 4 (BASELINE config 3's per-GPU workload), same box" 4 | tail -1
python scratch/pmc_traffic2.py $D/pmc_fetch $D/pmc_write 3 profiles/r05_hbm_traffic_pmc.json > /tmp/traffic.log 2>&1; tail -3 /tmp/traffic.log
python scratch/pmc_mfma_util.py $D/pmc_mfma > profiles/r05_${T}_pmc_sq_mfma_util_double_conv_256.csv
