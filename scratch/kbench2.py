import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
for (B,H,W,Ci,Co) in [(8,512,512,64,64),(8,128,128,256,256),(8,256,256,128,128)]:
    r = ops.bench_double_conv(B,H,W,Ci,Co,torch.bfloat16, iters=10)
    print(r['shape'], ' '.join(f"{k}:{v['ms']*1e3:.0f}us" for k,v in r.items() if isinstance(v, dict) and 'wgrad' not in k and 'all' not in k))
