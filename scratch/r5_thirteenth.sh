#!/bin/bash
# Slab reduce with split lanes that follow the split count (uh_slab16_lanes): the tests that exercise it, the per-launch durations of
# one step (kernel trace, one stream) and the step, interleaved with the round's previous library (scratch/libs/libunet_hip_prev.so).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r5t; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_wgrad_slabs.py tests/test_gpu_large.py tests/test_gpu_bf16_vs_reference.py -q -m gpu -x > $OUT/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained --no-b4-leg --no-strong-leg --no-side-stream"
for b in 8 4; do
  rocprofv3 --kernel-trace --stats -d $OUT/st$b -- $BENCH --batch $b --steps 10 --warmup 3 > $OUT/stats_b$b.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats_b$b.log; exit 1; }
  db=$(find $OUT/st$b -name "*.db" | head -1)
  python3 $R/scratch/rocpd_export.py stats $db $OUT/kernel_stats_b$b.csv; python3 $R/scratch/rocpd_export.py trace $db $OUT/kernel_trace_b$b.csv; rm -rf $OUT/st$b
done
cd $R
for i in 1 2 3; do
  UH_LIB_PATH=$R/scratch/libs/libunet_hip_prev.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg 2>$OUT/bb_$i.err > $OUT/bb_$i.json &&
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-inference --no-sustained --no-strong-leg 2>$OUT/bn_$i.err > $OUT/bn_$i.json || exit 1
done
python - <<'PY'
import json, csv
for b in (8, 4):
    rows = list(csv.DictReader(open(f"gpurun_out/r5t/kernel_trace_b{b}.csv")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sl = [r for r in rows if "slab_reduce_f16" in r["Kernel_Name"]][-17:]
    print("batch", b, "slab_reduce_f16pair, last step:", " ".join(f"{r['Grid_X']}:{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}" for r in sl),
          "| sum %.1f us" % sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in sl))
for i in (1, 2, 3):
    for t in ("bb", "bn"):
        j = [json.loads(l) for l in open(f"gpurun_out/r5t/{t}_{i}.json") if l.startswith("{")][0]
        print(t, i, j["value"], "img/s | b4", (j.get("per_gpu_batch4") or {}).get("value"))
PY
