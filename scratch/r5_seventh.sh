#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5g
python -m pytest tests/test_gpu_fused_tails.py tests/test_gpu_ops.py tests/test_gpu_bench_ranks.py tests/test_gpu_large.py -m gpu -q --no-header -p no:cacheprovider -k "upsample or up_ or Up or tail or one_rank or bilinear or 2gib" > gpurun_out/r5g/up_tests.log 2>&1; echo "up tests rc=$?"
tail -n 6 gpurun_out/r5g/up_tests.log
for i in 1 2; do
  python scratch/r4_upbench.py > gpurun_out/r5g/up_lds_$i.txt 2>&1; UH_UP_FWD_ROWS=1 python scratch/r4_upbench.py > gpurun_out/r5g/up_rows_$i.txt 2>&1
done
grep -h "fwd" gpurun_out/r5g/up_lds_1.txt gpurun_out/r5g/up_rows_1.txt gpurun_out/r5g/up_lds_2.txt gpurun_out/r5g/up_rows_2.txt | head -40
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg > gpurun_out/r5g/lds_$i.json 2> gpurun_out/r5g/lds_$i.err
  UH_UP_FWD_ROWS=1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg > gpurun_out/r5g/rows_$i.json 2> gpurun_out/r5g/rows_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5g/*.json")):
    j = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    print(f.split("/")[-1], j["value"], j["ms_per_step"], "b4", j["per_gpu_batch4"]["images_per_sec"])
PY
