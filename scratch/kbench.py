import sys, torch, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] != 'f32') else torch.float32
shapes = [(8,512,512,64,64),(8,256,256,64,128),(8,128,128,128,256),(8,64,64,256,512),(8,32,32,512,512),(8,64,64,1024,512),(8,128,128,512,256),(8,256,256,256,128),(8,512,512,128,64)]
tot = {}
for (B,H,W,Ci,Co) in shapes:
    r = ops.bench_double_conv(B,H,W,Ci,Co,dt, iters=10)
    print(r['shape'], ' '.join(f"{k}:{v['ms']*1e3:.0f}us/{v['tflops']:.0f}TF" for k,v in r.items() if isinstance(v, dict)))
