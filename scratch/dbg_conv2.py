import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, H, W, Ci, Co) in [(1, 448, 448, 64, 64), (4, 256, 256, 64, 64)]:
    x = torch.randn(B, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).bfloat16().float()
    ref = F.conv2d(x, w, padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    wf, _ = ops.pack_w3x3(w.to(dev), torch.bfloat16, False)
    y, stats, nslab = ops.conv3x3_fwd(xg, None, wf, Co, True)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    thr = 0.05 * float(ref.abs().max())
    print((B, H, W, Ci, Co), "rel err", float(err.max() / ref.abs().max()))
    ty, tx = H // 16, W // 16
    e = err.reshape(B, Co, ty, 16, tx, 16)
    bad_tiles = (e.amax(dim=(1, 3, 5)) > thr).nonzero().tolist()
    ntile = B * ty * tx
    ids = [b * ty * tx + yy * tx + xx for b, yy, xx in bad_tiles]
    print("  bad tiles:", len(ids), "of", ntile, "first ids", ids[:20], "min", min(ids) if ids else None)
    if ids:
        b, yy, xx = bad_tiles[0]
        t = e[b, :, yy, :, xx, :]                       # [Co, 16, 16]
        print("  in first bad tile: bad channels", (t.amax(dim=(1, 2)) > thr).nonzero().flatten().tolist())
        print("  bad rows", (t.amax(dim=(0, 2)) > thr).nonzero().flatten().tolist(), "bad cols", (t.amax(dim=(0, 1)) > thr).nonzero().flatten().tolist())
        gt = got.reshape(B, Co, ty, 16, tx, 16)[b, :, yy, :, xx, :]
        # does the bad tile equal the reference of ANOTHER tile?
        rt = ref.reshape(B, Co, ty, 16, tx, 16)
        d = (rt - gt[None, :, None, :, None, :]).abs().amax(dim=(1, 3, 5))       # [B, ty, tx]
        bb = d.flatten().argmin()
        print("  bad tile", (b, yy, xx), "best matches reference tile (flat id)", int(bb), "with diff", float(d.flatten()[bb]))
