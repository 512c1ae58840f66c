"""Join a rocprofv3 --pmc ... --kernel-trace CSV pair into one row per dispatch: duration, effective clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA-busy share, wait shares.   usage: python scratch/pmc_parse.py DIR [tag]"""
import collections, csv, glob, sys
d = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else d
cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
rows = collections.OrderedDict()
for f in cc:
    for r in csv.DictReader(open(f)):
        k = r["Dispatch_Id"]
        e = rows.setdefault(k, {"name": r["Kernel_Name"], "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for key in ("Start_Timestamp", "End_Timestamp"):
            if key in r and r[key]:
                e[key] = float(r[key])
for f in kt:
    for r in csv.DictReader(open(f)):
        k = r.get("Dispatch_Id")
        if k in rows:
            rows[k]["Start_Timestamp"] = float(r["Start_Timestamp"]); rows[k]["End_Timestamp"] = float(r["End_Timestamp"])
agg = collections.OrderedDict()
seen = collections.Counter()
for k, e in rows.items():
    n = e["name"]
    short = n.split("(")[0][:60]
    if "conv3x3" not in short:
        continue
    seen[short] += 1
    if "Start_Timestamp" not in e:
        continue
    dur = (e["End_Timestamp"] - e["Start_Timestamp"]) * 1e-3     # us
    c = e["c"]
    key = (short, round(dur / 8))                                   # group dispatches of one kernel by similar duration
    agg.setdefault(key, []).append((dur, c))
print(f"# {tag}")
print("kernel,n,dur_us,clock_ghz,mfma_busy_frac,wait_any,wait_inst_any,active_inst,busy_frac")
for (short, _), lst in agg.items():
    dur = sum(x[0] for x in lst) / len(lst)
    def avg(nm):
        v = [x[1].get(nm) for x in lst if nm in x[1]]
        return sum(v) / len(v) if v else float("nan")
    gui = avg("GRBM_GUI_ACTIVE"); wc = avg("SQ_WAVE_CYCLES"); mf = avg("SQ_VALU_MFMA_BUSY_CYCLES")
    clock = gui / 8 / (dur * 1e3) if gui == gui else float("nan")           # cycles per ns = GHz
    cyc = clock * dur * 1e3                                                   # shader cycles of the dispatch
    mfma = mf / 1024 / cyc if cyc == cyc else float("nan")                    # busy cycles summed over 1024 SIMDs
    wa, wi, ac = avg("SQ_WAIT_ANY"), avg("SQ_WAIT_INST_ANY"), avg("SQ_ACTIVE_INST_ANY")
    tot = wc if wc == wc and wc > 0 else float("nan")
    print(f"{short},{len(lst)},{dur:.1f},{clock:.3f},{mfma:.3f},{wa / tot:.3f},{wi / tot:.3f},{ac / tot:.3f},{avg('SQ_BUSY_CYCLES') / 8 / 4 / cyc if cyc == cyc else float('nan'):.3f}")
