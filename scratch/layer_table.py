"""Per-(kernel, grid) launch table from a rocprofv3 --kernel-trace CSV: which layers' conv launches are slow."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
pat = sys.argv[3] if len(sys.argv) > 3 else "conv3x3"
g = collections.OrderedDict()
for r in rows:
    if pat not in r["Kernel_Name"]:
        continue
    key = (re.sub(r"^_Z\d+", "", r["Kernel_Name"])[:34], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g.setdefault(key, []).append(d)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:36s} grid {k[1]:6d} x {k[2]:3d}  n/step {len(v)/nsteps:5.1f}  avg {sum(v)/len(v):8.1f} us  min {min(v):8.1f}  total/step {sum(v)/nsteps/1e3:7.3f} ms")
