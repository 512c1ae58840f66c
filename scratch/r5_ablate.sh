#!/bin/bash
# What each class of work costs inside the forward / backward-data conv kernel, per layer shape: scratch/r4_conv_bench.py on the default
# library and on the six diagnostic variants of scratch/libs/ (built here by scratch/r5_mklib.py abl_X -DUH_ABL_X=1; results of the
# variants are garbage by design).   scratch/r5_ablate.sh <outdir> [batch]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
for v in base NOW NODMA NOLDS NOMFMA NOSTORE NOSTATS base2; do
  if [ $v = base ] || [ $v = base2 ]; then unset UH_LIB_PATH; else export UH_LIB_PATH=$R/scratch/libs/libunet_hip_abl_$v.so; fi
  python scratch/r4_conv_bench.py ${2:-8} 2>&1 | grep -v amdgpu.ids > $OUT/$v.txt; echo "$v rc=$?"
done
python - <<PY
import re, glob
rows = {}
order = ["base", "base2", "NOW", "NODMA", "NOLDS", "NOMFMA", "NOSTORE", "NOSTATS"]
for v in order:
    for line in open("$OUT/%s.txt" % v):
        m = re.match(r"(\S+)\s+H=\s*(\d+)\s+(\d+) ->\s*(\d+) \| fwd\s+([\d.]+) us .*\| dgrad\s+([\d.]+) us", line)
        if m:
            rows.setdefault(m.group(1), {})[v] = (float(m.group(5)), float(m.group(6)))
print("forward us per launch; columns:", order)
for k, d in rows.items():
    print(f"{k:8s} fwd  ", " ".join(f"{d[v][0]:7.1f}" if v in d else "      -" for v in order))
    print(f"{k:8s} dgrad", " ".join(f"{d[v][1]:7.1f}" if v in d else "      -" for v in order))
PY
