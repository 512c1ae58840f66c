"""(round 3) Backward-weights from the inside: scratch/ab/lib_wgdiag.so (scratch/mk_conv_variant.py wgdiag) stamps s_memtime per wave at
kernel entry, behind the first tile fence, behind the tile loop and once the slab stores have drained.
usage: python scratch/diag_wgrad.py scratch/ab/lib_wgdiag.so"""
import ctypes, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratch.ab_conv import Lib
from unet_amd import _lib as L
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
lb = Lib(sys.argv[1])
lb.dll.uh_diag_set.argtypes = [ctypes.c_void_p]; lb.dll.uh_diag_set.restype = ctypes.c_int
for (B, H, W, Ci, Co) in [(8, 128, 128, 128, 256), (8, 128, 128, 256, 256), (8, 512, 512, 64, 64), (8, 64, 64, 1024, 512), (32, 128, 128, 128, 256)]:
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(B, H, W, Ci, generator=g)).to(dev, torch.bfloat16)
    dy = torch.randn(B, H, W, Co, generator=g).to(dev, torch.bfloat16)
    dw = torch.empty(Co * 9 * Ci, dtype=torch.float32, device=dev)
    nb = lb.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Ci, Co, L.UH_BF16)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dbg = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
    def run():
        lb.call("uh_conv3x3_wgrad", dy.data_ptr(), Co, x.data_ptr(), Ci, Ci, None, 0, 0, dw.data_ptr(), Co, ws.data_ptr(), nb, B, H, W, L.UH_BF16, st)
    lb.dll.uh_diag_set(None)
    for _ in range(30): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    lb.dll.uh_diag_set(ctypes.c_void_p(dbg.data_ptr()))
    run(); torch.cuda.synchronize()
    lb.dll.uh_diag_set(None)
    d = dbg.view(-1, 8).cpu().numpy()
    d = d[d[:, 0] == 4].astype(np.float64)
    total = d[:, 5] - d[:, 2]
    clk = total / (d[:, 1] * 10.0)
    pro, loop, slab = d[:, 3] - d[:, 2], d[:, 4] - d[:, 3], d[:, 5] - d[:, 4]
    tiles = d[:, 6]
    mf = 16.0 * 144 * tiles
    print(f"== backward-weights {Ci}x{Co} @{H}x{W} B{B}: {us:.1f} us per call (MFMA kernel + slab_reduce), {2.0 * B * H * W * Co * 9 * Ci / us / 1e6:.0f} TFLOP/s; "
          f"{len(d)} waves, {tiles.mean():.1f} tiles each; in-kernel clock {np.median(clk):.3f} GHz")
    q = lambda v: f"{v.mean():8.0f} ({v.mean() / total.mean() * 100:4.1f} %)  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}"
    print(f"   wave lifetime        {q(total)}   = {total.mean() / np.median(clk) / 1e3:.1f} us")
    print(f"   prologue             {q(pro)}")
    print(f"   tile loop            {q(loop)}   own MFMA issue {mf.mean():.0f} cycles = {mf.mean() / loop.mean():.2f} of it (x2 waves per SIMD = {2 * mf.mean() / loop.mean():.2f})")
    print(f"     of it tile fences  {q(d[:, 7])}   (s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier at the end of every tile)")
    print(f"   slab stores + drain  {q(slab)}   = {slab.mean() / np.median(clk) / 1e3:.1f} us")
