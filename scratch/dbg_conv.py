import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, H, W, Ci, Co) in [(1, 16, 16, 64, 64), (1, 32, 32, 64, 64), (1, 16, 16, 128, 64), (2, 48, 32, 64, 64)]:
    x = torch.randn(B, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).bfloat16().float()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    wf, _ = ops.pack_w3x3(w.to(dev), torch.bfloat16, False)
    y, stats, nslab = ops.conv3x3_fwd(xg, None, wf, Co, True)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2).double()
    err = (got - ref).abs()
    print((B, H, W, Ci, Co), "rel err", float(err.max() / ref.abs().max()))
    bad_c = (err.amax(dim=(0, 2, 3)) > 0.05 * ref.abs().max()).nonzero().flatten().tolist()
    bad_r = (err.amax(dim=(0, 1, 3)) > 0.05 * ref.abs().max()).nonzero().flatten().tolist()
    bad_x = (err.amax(dim=(0, 1, 2)) > 0.05 * ref.abs().max()).nonzero().flatten().tolist()
    print("  bad channels", bad_c[:64], "\n  bad rows", bad_r[:64], "\n  bad cols", bad_x[:64])
    if bad_c:
        # is a bad output a permutation of the reference channels / rows?
        c = bad_c[0]
        gg = got[0, c]
        best = min(((float((gg - ref[0, cc]).abs().max()), cc) for cc in range(Co)))
        print("  channel", c, "best matching ref channel", best)
        r = bad_r[0]
        best = min(((float((got[0, :, r] - ref[0, :, rr]).abs().max()), rr) for rr in range(H)))
        print("  row", r, "best matching ref row", best)
