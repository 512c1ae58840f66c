"""PMC probe: forward conv of the 64->64 512^2 layer, 6 launches with real stores (lib A) then 6 with dropped stores (lib B).
usage (under rocprofv3 --pmc ...): python3 scratch/pmc_store_probe.py A.so B.so"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratch.ab_conv import Lib
from unet_amd import _lib as L
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B, H, W, C = 8, 512, 512, 64
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(B, H, W, C, generator=g)).to(dev, torch.bfloat16)
w = (torch.randn(C, C, 3, 3, generator=g) / 24).to(dev)
for path in sys.argv[1:3]:
    lb = Lib(path)
    wf = torch.empty(C * 9 * C, dtype=torch.bfloat16, device=dev); wd = torch.empty_like(wf)
    lb.call("uh_pack_w3x3", w.data_ptr(), *w.stride(), C, C, wf.data_ptr(), wd.data_ptr(), L.UH_BF16 | 0x300, st)
    y = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dev)
    for _ in range(6):
        lb.call("uh_conv3x3_fwd", x.data_ptr(), C, C, None, 0, 0, wf.data_ptr(), y.data_ptr(), C, C, None, B, H, W, L.UH_BF16 | 0x100, st)
    torch.cuda.synchronize()
