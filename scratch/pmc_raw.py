"""Per-kernel averages of every counter in one or more rocprofv3 --pmc output dirs.  usage: python scratch/pmc_raw.py DIR [DIR ...]"""
import collections, csv, glob, sys
agg = collections.OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        rows = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "conv3x3" not in r["Kernel_Name"]:
                continue
            e = rows.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0][:44], "grid": r.get("Grid_Size"), "c": {}})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            e["dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
        for e in rows.values():
            key = (e["name"], e["grid"], int(round(e["dur"] / 10)))
            a = agg.setdefault(key, {"n": collections.Counter(), "s": collections.Counter(), "dur": [], })
            a["dur"].append(e["dur"])
            for k, v in e["c"].items():
                a["n"][k] += 1; a["s"][k] += v
for (name, grid, _), a in agg.items():
    dur = sum(a["dur"]) / len(a["dur"])
    print(f"{name} grid={grid} dur={dur:.1f}us n={len(a['dur'])}")
    for k in sorted(a["s"]):
        v = a["s"][k] / a["n"][k]
        print(f"    {k:36s} {v:16.0f}   per_us {v / dur:12.1f}")
