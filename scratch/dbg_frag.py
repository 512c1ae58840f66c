import os, sys, torch, numpy as np
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
dev = torch.device("cuda:0")

def wfrag_index(o, tap, k, K, mode, ES):
    CK = 64 // ES; VEC = 16 // ES
    if mode == 2:
        c32 = o & 31; g = ((o >> 5) << 1) | ((c32 >> 2) & 1); m = ((c32 >> 3) << 2) | (c32 & 3)
    elif mode == 1:
        c16 = o & 15; q = c16 >> 2; g = o >> 4; m = ((((q & 1) << 1) | (q >> 1)) << 2) | (c16 & 3)
    else:
        g = o >> 4; m = o & 15
    chunk = k // CK; e = k - chunk * CK; kp = e // VEC; v = e - kp * VEC
    return ((((g * 9 + tap) * (K // CK) + chunk) * 64 + (kp * 16 + m)) * VEC + v)

g = torch.Generator().manual_seed(0)
for (dtype, Cout, Cin, B, H, W) in [(torch.bfloat16, 128, 64, 2, 320, 336), (torch.float32, 64, 64, 1, 64, 64), (torch.bfloat16, 64, 64, 1, 64, 64)]:
    ES = 2 if dtype == torch.bfloat16 else 4
    mode = (2 if Cout % 128 == 0 else 1) if ES == 2 else 0
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    if ES == 2: w = w.bfloat16().float()
    wk, _ = ops.pack_w3x3(w.to(dev), dtype, True)
    wf, wd = ops.pack_w3x3(w.to(dev), dtype, True, None, True, True)
    torch.cuda.synchronize()
    wk_h = wk.float().cpu().numpy().reshape(Cout, 9, Cin)
    wf_h = wf.float().cpu().numpy()
    exp = np.zeros(Cout * 9 * Cin, np.float32)
    for o in range(0, Cout, 7):
        for t in range(9):
            idx = np.array([wfrag_index(o, t, k, Cin, mode, ES) for k in range(Cin)])
            exp[idx] = wk_h[o, t]
            if not np.array_equal(wf_h[idx], wk_h[o, t]):
                print("PACK MISMATCH", dtype, Cout, Cin, "row", o, "tap", t); break
    x = torch.randn(B, Cin, H, W, generator=g)
    if ES == 2: x = x.bfloat16().float()
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    y1, _, _ = ops.conv3x3_fwd(xg, None, wk, Cout, False)
    y2, _, _ = ops.conv3x3_fwd(xg, None, wf, Cout, False, None, True)
    ref = F.conv2d(x, w, padding=1).permute(0, 2, 3, 1)
    e1 = float((y1.float().cpu() - ref).abs().max() / ref.abs().max()); e2 = float((y2.float().cpu() - ref).abs().max() / ref.abs().max())
    print(dtype, Cout, Cin, "krsc err", e1, "frag err", e2)
    if e2 > 0.05:
        d = (y2.float().cpu() - ref).abs().amax(dim=(0, 1, 2))
        print("  bad channels:", (d > 0.05 * float(ref.abs().max())).nonzero().flatten().tolist()[:64])
