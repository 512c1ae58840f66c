import collections, csv, glob, sys
for d in sys.argv[1:]:
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "conv3x3_fwd_mfma_v2" not in r["Kernel_Name"]:
            continue
        e = rows.setdefault(r["Dispatch_Id"], {"c": collections.OrderedDict()})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
    rows = list(rows.values())
    n = len(rows) // 2
    for label, grp in (("real stores", rows[1:n]), ("dropped stores", rows[n + 1:])):
        print(f"{d} {label}: dur {sum(e['dur'] for e in grp) / len(grp):.1f} us  " +
              "  ".join(f"{k}={sum(e['c'][k] for e in grp) / len(grp):.4g}" for k in grp[0]["c"]))
