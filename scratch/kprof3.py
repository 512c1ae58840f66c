import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
r = ops.bench_double_conv(8, 512, 512, 64, 64, torch.bfloat16, iters=3)
print(r)
