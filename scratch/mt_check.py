"""multi-tile (persistent workgroup walks > 1 tile) correctness of the forward conv against torch fp64 on CPU, small enough
for the CPU: B=1, 448x448 -> 784 tiles > 768 workgroups (NBW=1) ; and 128-channel output (NBW=2: 784*1 slabs >= 512)."""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (B, H, W, Ci, Co) in [(1, 448, 448, 64, 64), (1, 448, 448, 32, 128), (2, 320, 336, 64, 128)]:
    x = torch.randn(B, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    wf, _ = ops.pack_w3x3(w.to(dev), torch.bfloat16, False)
    y, stats, nslab = ops.conv3x3_fwd(xg, None, wf, Co, True)
    torch.cuda.synchronize()
    err = ((y.float().cpu().permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()).item()
    print((B, H, W, Ci, Co), 'rel err', err, flush=True)
    assert err < 1e-2
print('OK')
