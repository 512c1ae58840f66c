#!/bin/bash
# Which kernels of the train step are bound by VALU issue rather than HBM?  SQ_INSTS_VALU per launch over two steps of bench.py:
# VALU wave-instructions / (1024 SIMDs) x 4 cycles against the launch's duration at 2.1 GHz.   scratch/r4_valu_pmc.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/p1 -- python3 $R/bench.py --no-cpu-baseline --no-kernel-profile --no-inference --no-sustained --no-b4-leg --no-side-stream --steps 2 --warmup 1 > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
db=$(find $OUT/p1 -name "*.db" | head -1); mkdir -p $OUT/c1; python3 $R/scratch/rocpd_export.py counters $db $OUT/c1; rm -rf $OUT/p1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/c1/export_counter_collection.csv")))
d = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"][:46]
    e = d.setdefault(k, collections.defaultdict(list))
    e[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        e["dur"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = []
for k, e in d.items():
    n = len(e["dur"])
    valu, dur, act = sum(e["SQ_INSTS_VALU"]), sum(e["dur"]), sum(e["GRBM_GUI_ACTIVE"])
    if not dur: continue
    cyc = act / max(n, 1)                      # GPU cycles per launch
    frac = (valu / n / 1024 * 4) / cyc if cyc else 0
    out.append((dur, k, n, dur / n, valu / n / 1e6, frac))
print(f"{'kernel':46s} {'calls':>5s} {'us':>8s} {'M valu':>8s}  VALU issue / cycles")
for dur, k, n, us, mv, frac in sorted(out, reverse=True)[:32]:
    print(f"{k:46s} {n:5d} {us:8.1f} {mv:8.2f}  {frac:5.2f}")
PY
