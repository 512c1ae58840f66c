"""Overlap report of one traced train step: time with 0 / 1 / >= 2 kernels running, and the kernel pairs that overlap most.
    python scratch/r4_overlap.py <kernel_trace.csv> <steps in the trace>"""
import csv, sys
from collections import Counter
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steps are delimited by the optimizer kernel
ends = [i for i, e in enumerate(ev) if "rmsprop_kernel" in e[2]]
a, b = ends[len(ends) // 2 - 1] + 1, ends[len(ends) // 2] + 1
seg = ev[a:b]
tot = sum(e - s for s, e, _ in seg)
wall = max(e for _, e, _ in seg) - seg[0][0]
pts = sorted([(s, 1) for s, _, _ in seg] + [(e, -1) for _, e, _ in seg])
cur, last, cov = 0, pts[0][0], [0, 0, 0]
for t, d in pts:
    cov[min(cur, 2)] += t - last
    cur += d
    last = t
print(f"launches {len(seg)}  sum of durations {tot / 1e6:.3f} ms  wall {wall / 1e6:.3f} ms  idle {cov[0] / 1e6:.3f}  one kernel {cov[1] / 1e6:.3f}  two or more {cov[2] / 1e6:.3f}")


def short(n):
    for k in ("conv3x3_wgrad_mfma_v2", "conv3x3_fwd_mfma_v2", "slab_reduce", "bn_relu_pool_bwd_apply", "bn_relu_pool_bwd_reduce", "bn_relu_bwd_apply",
              "bn_relu_bwd_reduce", "bn_bwd_finalize", "upsample2x_bwd", "bn_relu_head_bwd", "stem_mfma"):
        if k in n:
            return k
    return n[:30]


c = Counter()
for i, (s, e, nm) in enumerate(seg):
    for j in range(i + 1, len(seg)):
        s2, e2, nm2 = seg[j]
        if s2 >= e:
            break
        ov = min(e, e2) - s2
        if ov > 0:
            c[tuple(sorted((short(nm), short(nm2))))] += ov
for k, v in c.most_common(14):
    print(f"{v / 1e3:8.1f} us  {k[0]}  ||  {k[1]}")
# per backward-weights launch: duration, and how much of it ran beside something else
print("backward-weights launches in order: duration us (overlapped us)")
line = []
for i, (s, e, nm) in enumerate(seg):
    if "wgrad_mfma" not in nm:
        continue
    ov = 0
    for j, (s2, e2, nm2) in enumerate(seg):
        if j != i and s2 < e and e2 > s and "wgrad" not in nm2 and "slab_reduce" not in nm2:
            ov += min(e, e2) - max(s, s2)
    line.append(f"{(e - s) / 1e3:.0f}({ov / 1e3:.0f})")
print(" ".join(line))
