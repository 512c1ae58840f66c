"""One train step's dispatch sequence from two kernel traces side by side (same launch order, e.g. batch 4 vs batch 8).
    python scratch/r4_seq.py trace_a.csv trace_b.csv [filter-substring]"""
import csv, re, sys
def load(p):
    rows = list(csv.DictReader(open(p)))
    # steps are delimited by the pack kernel
    idx = [i for i, r in enumerate(rows) if 'pack_w3x3_batched' in r['Kernel_Name']]
    a, b = idx[-3], idx[-2]
    return rows[a:b]
A, B = load(sys.argv[1]), load(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ''
assert len(A) == len(B), (len(A), len(B))
ta = tb = 0
t0a, t0b = int(A[0]['Start_Timestamp']), int(B[0]['Start_Timestamp'])
for x, y in zip(A, B):
    n = re.sub(r'^_Z\d+', '', x['Kernel_Name']); n = re.sub(r'^void ', '', n)[:46]
    da, db = int(x['DurationNs']) / 1e3, int(y['DurationNs']) / 1e3
    if flt and flt not in x['Kernel_Name']: continue
    ta += da; tb += db
    print(f"{n:46s} grid {x['Grid_X']:>8s}/{y['Grid_X']:>8s} wg {x['Workgroup_X']:>4s} {da:8.1f} {db:8.1f} ratio {da/db:5.2f}  t={ (int(x['Start_Timestamp'])-t0a)/1e3:8.1f} {(int(y['Start_Timestamp'])-t0b)/1e3:8.1f}")
print(ta, tb)
print("span", (int(A[-1]['End_Timestamp'])-t0a)/1e3, (int(B[-1]['End_Timestamp'])-t0b)/1e3)
