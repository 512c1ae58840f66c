"""How the bf16 path is judged: against the reference's OWN bf16.

For a quantity q the yardstick is   own(q) = err(reference under torch.autocast('cpu', bfloat16), reference fp32)
-- train.py:116, and AMP is the reference CLI's default (train.py:233) -- taken either from fixture set G15 (the reference's own
modules, tests/golden/make_golden.py) or, for shapes no fixture holds, from oracle/nn_ref.py on the same inputs (the same stock
torch.nn graph, bit-identical to G15 in both legs: tests/test_oracle_golden.py).  The HIP bf16 path is held to

        err(HIP bf16, reference fp32)  <=  FACTOR * own(q)            (relative L2 for tensors, relative difference for scalars)

with FACTOR = 1.5 for a tensor of >= 1024 elements of a single forward / backward pass; 2.5 for smaller tensors (a BatchNorm
weight of 8 elements is eight coin flips) and for what sits behind RMSprop's sign-like first steps (later steps of a trajectory,
parameter travel: a weight whose tiny gradient changes sign moves the other way by 20 lr, in the reference's bf16 run as in ours);
scalars get a floor of one bf16 ulp (2^-8): the reference's bf16 value of a scalar can land on its fp32 value by luck (bce of G15's
full UNet: 2e-6), which is not a precision anybody has.  Over ALL rows of a test the median ratio must stay <= 1.15: the HIP path
as a whole is as close to fp32 as the reference's bf16 is (measured: 1.00-1.02).
Every row goes to gpurun_out/bf16_vs_reference_report.txt."""
import os
import statistics

import numpy as np
import torch

ULP = 2.0 ** -8
FACTOR, FACTOR_LOOSE = 1.5, 2.5
MEDIAN_MAX = 1.15
ROWS = []


def l2(a, b):
    a = np.asarray(a.detach().double().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().double().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def dump():
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bf16_vs_reference_report.txt", "w") as f:
        f.write(f"{'quantity':62s} {'HIP bf16':>10s} {'ref bf16':>10s} {'ratio':>6s}   (both against the reference's fp32 result)\n")
        f.write("\n".join(ROWS) + "\n")


class Collector:
    def __init__(self, tag):
        self.tag, self.bad, self.ratios = tag, [], []

    def add(self, what, hip_err, own_err, floor=0.0, factor=FACTOR):
        bound = max(factor * own_err, floor)
        ratio = hip_err / max(own_err, 1e-300)
        if own_err > floor / FACTOR:                       # rows decided by their floor say nothing about the ratio
            self.ratios.append(ratio)
        ROWS.append(f"{self.tag + ' ' + what:62s} {hip_err:10.3e} {own_err:10.3e} {ratio:6.2f}"
                    + ("" if hip_err <= bound else f"   > bound {bound:.3e}"))
        if hip_err > bound:
            self.bad.append(f"{what}: HIP bf16 {hip_err:.3e} vs the fp32 reference; the reference's own bf16 {own_err:.3e} (bound {bound:.3e})")

    def tensor(self, what, hip, ref32, ref16, floor=0.0, loose=False):
        n = int(np.asarray(ref32.detach().cpu() if torch.is_tensor(ref32) else ref32).size)
        self.add(what, l2(hip, ref32), l2(ref16, ref32), floor, FACTOR_LOOSE if (loose or n < 1024) else FACTOR)

    def scalar(self, what, hip, ref32, ref16, loose=False, floor=None):
        r = float(ref32)
        self.add(what, abs(float(hip) - r) / abs(r), abs(float(ref16) - r) / abs(r), ULP if floor is None else floor,
                 FACTOR_LOOSE if loose else FACTOR)

    def done(self):
        dump()
        assert not self.bad, "\n".join(self.bad)
        if len(self.ratios) >= 5:
            med = statistics.median(self.ratios)
            assert med <= MEDIAN_MAX, f"{self.tag}: median err(HIP bf16) / err(reference bf16) over {len(self.ratios)} quantities = {med:.3f}"
