"""Kernel-trace overlap: how much of each backward-weights launch runs while another kernel is also running.
usage: python scratch/overlap_trace.py DIR"""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[len(rows) * 2 // 3][0]
rows = [r for r in rows if r[0] >= t0]
wg = [r for r in rows if "wgrad_mfma" in r[2]]
oth = [r for r in rows if "wgrad_mfma" not in r[2]]
tot = ov = 0
for s, e, n, q, st in wg[:40]:
    o = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, n2, _, _ in oth if e2 > s and s2 < e)
    tot += e - s; ov += o
print(f"{len(wg)} wgrad launches in the window; overlapped fraction of the first 40: {ov / max(tot, 1):.3f}; queues {sorted(set(r[3] for r in rows))} streams {sorted(set(r[4] for r in rows))}")
# a stretch of the timeline
s0 = wg[5][0]
for s, e, n, q, st in rows:
    if s0 - 200000 < s < s0 + 600000:
        print(f"{(s - s0) / 1e3:9.1f} {(e - s0) / 1e3:9.1f} q{q} s{st} {n[:60]}")
