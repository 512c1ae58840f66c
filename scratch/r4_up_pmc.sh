#!/bin/bash
# What is the bilinear forward waiting for?  SQ counters over scratch/r4_upbench.py (two --pmc passes).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/p1 -- python3 $R/scratch/r4_upbench.py > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
db=$(find $OUT/p1 -name "*.db" | head -1); mkdir -p $OUT/c1; python3 $R/scratch/rocpd_export.py counters $db $OUT/c1; rm -rf $OUT/p1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVES SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/p2 -- python3 $R/scratch/r4_upbench.py > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; echo pass2 failed; }
db=$(find $OUT/p2 -name "*.db" | head -1); mkdir -p $OUT/c2; [ -n "$db" ] && python3 $R/scratch/rocpd_export.py counters $db $OUT/c2; rm -rf $OUT/p2
python3 - <<PY
import csv, collections
for c in ("c1", "c2"):
    try:
        rows = list(csv.DictReader(open("$OUT/%s/export_counter_collection.csv" % c)))
    except Exception as e:
        print(c, "no data", e); continue
    d = collections.OrderedDict()
    for r in rows:
        if "upsample" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"][:40], r["Grid_Size"])
        e = d.setdefault(k, collections.defaultdict(list))
        e[r["Counter_Name"]].append(float(r["Counter_Value"]))
        e["dur"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for k, e in d.items():
        print(c, k, {n: round(sum(v) / len(v), 1) for n, v in e.items()})
PY
