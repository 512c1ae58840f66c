"""Side by side per-kernel totals of two kernel_stats csv files (ms per step), e.g. batch 4 against batch 8.
    python scratch/r4_cmp_stats.py a.csv b.csv [steps=13]"""
import csv, re, sys
a, b = sys.argv[1], sys.argv[2]
n = float(sys.argv[3]) if len(sys.argv) > 3 else 13
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']) / n, float(r['TotalDurationNs']) / 1e6 / n, float(r['AverageNs']) / 1e3)
    return d
A, B = load(a), load(b)
names = sorted(set(A) | set(B), key=lambda k: -(B.get(k, (0, 0, 0))[1]))
ta = tb = 0
print(f"{'kernel':64s} {'calls':>5s} {'a ms':>7s} {'a us':>7s} | {'b ms':>7s} {'b us':>7s} | a/b")
for k in names:
    x, y = A.get(k, (0, 0, 0)), B.get(k, (0, 0, 0))
    ta += x[1]; tb += y[1]
    short = re.sub(r'^_Z\d+', '', k)[:64]
    print(f"{short:64s} {y[0]:5.1f} {x[1]:7.3f} {x[2]:7.1f} | {y[1]:7.3f} {y[2]:7.1f} | {x[1] / y[1] if y[1] else 0:5.2f}")
print("sum ms/step", round(ta, 3), round(tb, 3))
