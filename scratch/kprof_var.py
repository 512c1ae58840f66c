"""Run the six conv kernels of a DoubleConv a few times for a PMC pass (library chosen by UH_LIB_PATH).
usage: python scratch/kprof_var.py [shape ...]   shapes as in scratch/ab_conv.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
SH = {"down2": (8, 128, 128, 128, 256), "l64": (8, 512, 512, 64, 64), "down1": (8, 256, 256, 64, 128)}
for name in (sys.argv[1:] or ["down2"]):
    B, H, W, Ci, Co = SH[name]
    r = ops.bench_double_conv(B, H, W, Ci, Co, torch.bfloat16, iters=3)
    print(name, {k: v for k, v in r.items() if k != "shape"})
