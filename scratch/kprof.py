import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
dt = torch.bfloat16
B,H,W,Ci,Co = 8,128,128,256,256
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(B,H,W,Ci,generator=g).to(dev, dt)
w = (torch.randn(Co,Ci,3,3,generator=g)/48).to(dev)
wf, wd = ops.pack_w3x3(w, dt, True)
dw = torch.empty(Co*9*Ci, dtype=torch.float32, device=dev)
for _ in range(3):
    y,_,_ = ops.conv3x3_fwd(x, None, wf, Co, True)
    ops.conv3x3_wgrad(x, x, None, dw)
torch.cuda.synchronize()
