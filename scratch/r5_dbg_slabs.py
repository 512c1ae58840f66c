"""Debug: block-scaled fp16 slabs.  Calls uh_conv3x3_wgrad with a caller-owned workspace, then decodes the slabs + inverse scales
on the host (numpy) and compares (a) the host decode with fp64, (b) the device reduce with fp64."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
from unet_amd._lib import LIB, UH_BF16
dev = torch.device("cuda:0")
for (B, H, W, Cin, Cout) in [(2, 24, 40, 64, 128), (2, 24, 40, 128, 128), (2, 64, 64, 64, 64), (2, 128, 128, 128, 256)]:
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g)).bfloat16().float()
    dy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    wd = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x.double(), wd, padding=1), [wd], dy.double())
    ref = ref.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).numpy()                      # KRSC
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    nbytes = LIB.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Cin, Cout, UH_BF16)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    dw = torch.zeros(Cout * 9 * Cin, dtype=torch.float32, device=dev)
    LIB.call("uh_conv3x3_wgrad", dyg.data_ptr(), Cout, xg.data_ptr(), Cin, Cin, None, 0, 0, dw.data_ptr(), Cout, ws.data_ptr(), nbytes,
             B, H, W, UH_BF16, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dev_out = dw.cpu().numpy().reshape(Cout, 9 * Cin)
    # plan (mirrors wgrad_plan): NWR 4 when Cout % 128 == 0
    nwr = 4 if Cout % 128 == 0 else 2
    coblk = 32 * nwr
    ctiles = (Cin // 64) * (Cout // coblk)
    ntile = B * ((W + 15) // 16) * ((H + 7) // 8)
    total = 256 if nwr == 4 else 512
    nsplit = max(1, min((total + ctiles - 1) // ctiles, ntile))
    npair = Cout // 2 * 9 * Cin
    raw = ws.cpu().numpy()
    slabs = raw[:nsplit * npair * 4].view(np.uint32).reshape(nsplit, Cout // 2, 9, Cin)
    inv = raw[nsplit * npair * 4: nsplit * npair * 4 + nsplit * ctiles * 4].view(np.float32).reshape(nsplit, ctiles)
    lo = (slabs & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
    hi = (slabs >> 16).astype(np.uint16).view(np.float16).astype(np.float64)
    nci = Cin // 64
    host = np.zeros((Cout, 9, Cin))
    for k in range(nsplit):
        for cp in range(Cout // 2):
            for cit in range(nci):
                s = inv[k, ((2 * cp) // coblk) * nci + cit]
                host[2 * cp, :, cit * 64:(cit + 1) * 64] += lo[k, cp, :, cit * 64:(cit + 1) * 64] * s
                host[2 * cp + 1, :, cit * 64:(cit + 1) * 64] += hi[k, cp, :, cit * 64:(cit + 1) * 64] * s
    host = host.reshape(Cout, 9 * Cin)
    rel = lambda a: np.linalg.norm(a - ref) / np.linalg.norm(ref)
    print(f"B{B} {H}x{W} {Cin}->{Cout} nsplit {nsplit} ctiles {ctiles}: host decode vs fp64 {rel(host):.3e}   device reduce vs fp64 {rel(dev_out):.3e}   "
          f"inv scales min {inv.min():.3e} max {inv.max():.3e} finite {np.isfinite(inv).all()}   |lo| max {np.abs(lo).max():.1f}", flush=True)
    if rel(dev_out) > 1e-2:
        r = dev_out / np.where(np.abs(host) > 1e-3 * np.abs(host).max(), host, np.nan)
        print("   device / host-decode ratio: median", np.nanmedian(r), "p10", np.nanpercentile(r, 10), "p90", np.nanpercentile(r, 90))
        bad = np.abs(dev_out - host) > 1e-2 * np.abs(host).max()
        rows = np.where(bad.any(1))[0]
        print("   rows with mismatches:", rows[:20], "count", len(rows), "of", Cout, "; cols of row", rows[0] if len(rows) else None, np.where(bad[rows[0]])[0][:20] if len(rows) else None)
