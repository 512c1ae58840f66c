"""Interleaved A/B timing of conv kernel variants in ONE process (cdna_hip_programming.md rule 24).

    python scratch/ab_conv.py [--rounds R] [--iters N] [--shapes down2,l64,...] name=path.so [name=path.so ...]

Every library is a build of the same C ABI (include/unet_hip.h; scratch/mkvariant.py makes them).  For each shape the
conv kernels are called back to back (`iters` launches between two events) round-robin over the variants, R rounds;
reports median / min ms and TFLOP/s per (variant, kernel).  Results of variant 0 are the correctness reference: max
relative difference of the outputs is printed (ablation builds are wrong by construction)."""
import argparse
import ctypes
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unet_amd import _lib as L  # noqa: E402

SHAPES = {
    # name: (B, H, W, Cin, Cout)   DoubleConv(Cin -> Cout -> Cout)
    "down2": (8, 128, 128, 128, 256),
    "l64": (8, 512, 512, 64, 64),
    "down1": (8, 256, 256, 64, 128),
    "down3": (8, 64, 64, 256, 512),
    "down4": (8, 32, 32, 512, 512),
    "up1": (8, 64, 64, 1024, 256),
    "up3": (8, 256, 256, 256, 64),
}


class Lib:
    def __init__(self, path):
        self.dll = ctypes.CDLL(path)
        for name, (ret, types) in L.parse_header().items():
            if not hasattr(self.dll, name):       # an older build of the ABI (A/B baselines)
                continue
            fn = getattr(self.dll, name)
            fn.restype = L._RET[ret]
            fn.argtypes = [ctypes.c_void_p if t == "ptr" else L._CTYPES[t] for t in types]

    def call(self, name, *a):
        rc = getattr(self.dll, name)(*a)
        if rc != 0:
            raise RuntimeError(f"{name}: {self.dll.uh_last_error().decode()}")

    def query(self, name, *a):
        return getattr(self.dll, name)(*a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shapes", default="down2")
    ap.add_argument("--kernels", default="fwd_conv1,fwd_conv2,dgrad_conv2,dgrad_conv1,wgrad_conv2,wgrad_conv1")
    ap.add_argument("--json", default="")
    ap.add_argument("--krsc", action="store_true", help="KRSC filter packs for every library (no fragment-major packs)")
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    libs = [(a.split("=")[0], Lib(a.split("=")[1])) for a in args.libs]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    dt = L.UH_BF16
    want = args.kernels.split(",")
    report = {}
    for sname in args.shapes.split(","):
        B, H, W, Cin, Cout = SHAPES[sname]
        g = torch.Generator().manual_seed(0)
        x = torch.relu(torch.randn(B, H, W, Cin, generator=g)).to(dev, torch.bfloat16)
        h = torch.relu(torch.randn(B, H, W, Cout, generator=g)).to(dev, torch.bfloat16)
        dyv = torch.randn(B, H, W, Cout, generator=g).to(dev, torch.bfloat16)
        w1 = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
        w2 = (torch.randn(Cout, Cout, 3, 3, generator=g) / (3 * Cout ** 0.5)).to(dev)
        base = libs[0][1]

        def pack(lb, w, flags=0):
            O, I = w.shape[0], w.shape[1]
            wf = torch.empty(O * 9 * I, dtype=torch.bfloat16, device=dev)
            wd = torch.empty(O * 9 * I, dtype=torch.bfloat16, device=dev)
            sO, sI, sH, sW = w.stride()
            lb.call("uh_pack_w3x3", w.data_ptr(), sO, sI, sH, sW, O, I, wf.data_ptr(), wd.data_ptr(), dt | flags, st)
            return wf, wd
        w1f, w1d = pack(base, w1)
        w2f, w2d = pack(base, w2)
        # fragment-major packs (UH_WFRAG = 0x100 forward copy, 0x200 backward-data copy) for the libraries that know them
        fragp = {}
        for ln, lb in libs:
            if hasattr(lb.dll, "uh_conv3x3_wfrag_ok") and not args.krsc:
                fragp[ln] = (pack(lb, w1, 0x300), pack(lb, w2, 0x300))
        yo = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
        xo = torch.empty(B, H, W, Cin, dtype=torch.bfloat16, device=dev)
        nslab = base.query("uh_conv3x3_stat_slabs", B, H, W, Cin, Cout, dt)
        stats = torch.empty(nslab * (2 * Cout + 2), dtype=torch.float32, device=dev)
        dw = torch.empty(Cout * 9 * max(Cin, Cout), dtype=torch.float32, device=dev)      # (conv1's result is Cout x 9 x Cin: larger than conv2's when Cin > Cout)
        wsb = max(base.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Cout, Cout, dt), base.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Cin, Cout, dt))
        # every variant may plan a different workspace: take the largest
        for _, lb in libs:
            wsb = max(wsb, lb.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Cout, Cout, dt), lb.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Cin, Cout, dt))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

        cur = {"name": None}

        def fwd(lb, src, cin, wp, dst, cout, stat):
            flag = 0
            fp = fragp.get(cur["name"])
            if fp is not None:          # swap the KRSC pack for this library's fragment-major copy of the same filter
                wp = {id(w1f): fp[0][0], id(w1d): fp[0][1], id(w2f): fp[1][0], id(w2d): fp[1][1]}[id(wp)]
                flag = 0x100
            lb.call("uh_conv3x3_fwd", src.data_ptr(), cin, cin, None, 0, 0, wp.data_ptr(), dst.data_ptr(), cout, cout,
                    None if stat is None else stat.data_ptr(), B, H, W, dt | flag, st)

        def wgrad(lb, dy, src, cin):
            lb.call("uh_conv3x3_wgrad", dy.data_ptr(), Cout, src.data_ptr(), cin, cin, None, 0, 0, dw.data_ptr(), Cout,
                    ws.data_ptr(), wsb, B, H, W, dt, st)

        f1 = 2.0 * B * H * W * Cout * 9 * Cin
        f2 = 2.0 * B * H * W * Cout * 9 * Cout
        kernels = {
            "fwd_conv1": (lambda lb: fwd(lb, x, Cin, w1f, yo, Cout, stats), f1, lambda: yo),
            "fwd_conv2": (lambda lb: fwd(lb, h, Cout, w2f, yo, Cout, stats), f2, lambda: yo),
            "dgrad_conv2": (lambda lb: fwd(lb, dyv, Cout, w2d, yo, Cout, None), f2, lambda: yo),
            "dgrad_conv1": (lambda lb: fwd(lb, dyv, Cout, w1d, xo, Cin, None), f1, lambda: xo),
            "wgrad_conv2": (lambda lb: wgrad(lb, dyv, h, Cout), f2, lambda: dw),
            "wgrad_conv1": (lambda lb: wgrad(lb, dyv, x, Cin), f1, lambda: dw[:Cout * 9 * Cin]),
        }
        kernels = {k: v for k, v in kernels.items() if k in want}
        times = {(ln, k): [] for ln, _ in libs for k in kernels}
        diffs = {}
        # correctness vs variant 0
        def run(ln, lb, fn):
            cur["name"] = ln
            fn(lb)
        for k, (fn, _, out) in kernels.items():
            run(libs[0][0], libs[0][1], fn); torch.cuda.synchronize()
            ref = out().float().clone()
            for ln, lb in libs[1:]:
                out().zero_()
                run(ln, lb, fn); torch.cuda.synchronize()
                diffs[(ln, k)] = float((out().float() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        # warm the clocks
        for _ in range(3):
            for ln, lb in libs:
                for k, (fn, _, _) in kernels.items():
                    run(ln, lb, fn)
        torch.cuda.synchronize()
        for r in range(args.rounds):
            order = libs if r % 2 == 0 else libs[::-1]
            for ln, lb in order:
                for k, (fn, _, _) in kernels.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        run(ln, lb, fn)
                    e1.record()
                    torch.cuda.synchronize()
                    times[(ln, k)].append(e0.elapsed_time(e1) / args.iters)
        print(f"== {sname}: B{B} {H}x{W} {Cin}->{Cout}->{Cout} bf16, {args.rounds} rounds x {args.iters} launches", flush=True)
        hdr = f"{'kernel':14s}" + "".join(f"{ln:>22s}" for ln, _ in libs)
        print(hdr)
        for k, (_, fl, _) in kernels.items():
            row = f"{k:14s}"
            for ln, _ in libs:
                t = times[(ln, k)]
                med, mn = statistics.median(t), min(t)
                row += f"  {med * 1e3:7.1f}us {fl / med / 1e9:6.0f}TF {diffs.get((ln, k), 0.0):.0e}"
                report[f"{sname}.{k}.{ln}"] = {"med_us": med * 1e3, "min_us": mn * 1e3, "tflops_med": fl / med / 1e9, "diff": diffs.get((ln, k), 0.0)}
            print(row, flush=True)
        tot = {ln: sum(statistics.median(times[(ln, k)]) for k in kernels) for ln, _ in libs}
        ftot = sum(v[1] for v in kernels.values())
        print(f"{'all':14s}" + "".join(f"  {tot[ln] * 1e3:7.1f}us {ftot / tot[ln] / 1e9:6.0f}TF      " for ln, _ in libs), flush=True)
    if args.json:
        json.dump(report, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
