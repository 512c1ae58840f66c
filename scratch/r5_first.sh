#!/bin/bash
# round 5, first GPU call: new parity tests (bf16 against the reference's own bf16, UNet_S fixture, PRE library in-session),
# the slab-precision experiment, and a baseline bench line
set -o pipefail
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_bf16_vs_reference.py -m gpu -q -x --no-header -p no:cacheprovider > gpurun_out/r5a/bf16_tests.log 2>&1; echo "bf16 tests rc=$?"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pre_fusion.py tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r5a/parity_tests.log 2>&1; echo "parity tests rc=$?"
rm -rf /tmp/r5slab
UH_WGRAD_SLAB_F32=1 python scratch/r5_slab_structured.py f32_strided --steps 40 > gpurun_out/r5a/slab_f32_strided.log 2>&1 && \
UH_WGRAD_SLAB_F32=1 UH_WGRAD_CONTIG=1 python scratch/r5_slab_structured.py f32_contig > gpurun_out/r5a/slab_f32_contig.log 2>&1 && \
python scratch/r5_slab_structured.py s16_strided > gpurun_out/r5a/slab_s16_strided.log 2>&1 && \
UH_WGRAD_CONTIG=1 python scratch/r5_slab_structured.py s16_contig > gpurun_out/r5a/slab_s16_contig.log 2>&1 && \
python scratch/r5_slab_cmp.py > gpurun_out/r5a/slab_cmp.log 2>&1; echo "slab rc=$?"
python bench.py --no-cpu-baseline > gpurun_out/r5a/bench_strided.json 2> gpurun_out/r5a/bench_strided.err; echo "bench rc=$?"
UH_WGRAD_CONTIG=1 python bench.py --no-cpu-baseline > gpurun_out/r5a/bench_contig.json 2> gpurun_out/r5a/bench_contig.err; echo "bench rc=$?"
python bench.py --no-cpu-baseline > gpurun_out/r5a/bench_strided2.json 2> gpurun_out/r5a/bench_strided2.err; echo "bench rc=$?"
UH_WGRAD_CONTIG=1 python bench.py --no-cpu-baseline > gpurun_out/r5a/bench_contig2.json 2> gpurun_out/r5a/bench_contig2.err; echo "bench rc=$?"
tail -3 gpurun_out/r5a/bf16_tests.log gpurun_out/r5a/parity_tests.log
