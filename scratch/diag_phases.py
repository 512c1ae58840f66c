"""(round 3) Where the time of a forward conv launch goes, per wave: a diagnostic build of conv3x3_fwd_mfma_v2 (scratch/ab/lib_diag.so, made by
patching s_memtime stamps into a copy of csrc/conv3x3.hip: kernel entry, after the prologue fence, behind the third column shift of
every chunk, behind every chunk fence, in front of / behind the stores, end of tile) is launched on the two forward shapes of the
256-channel DoubleConv; prints the mean core cycles per phase over all waves.  usage: python scratch/diag_phases.py scratch/ab/lib_diag.so [scratch/ab/lib_diagmin.so]
(the second library stamps kernel entry and exit only: the in-kernel clock and wave lifetimes of the UNPERTURBED schedule)."""
import ctypes, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratch.ab_conv import Lib
from unet_amd import _lib as L
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
lb = Lib(sys.argv[1])
lb.dll.uh_diag_set.argtypes = [ctypes.c_void_p]; lb.dll.uh_diag_set.restype = ctypes.c_int
SHAPES = [(8, 128, 128, 128, 256, True), (8, 128, 128, 256, 256, True), (8, 128, 128, 256, 256, False)]
MIN_SHAPES = SHAPES + [(8, 64, 64, 1024, 512, True), (8, 512, 512, 64, 64, True), (8, 512, 512, 64, 64, False), (8, 256, 256, 128, 128, True), (32, 128, 128, 128, 256, True)]
jobs = [(lb, sh, True) for sh in SHAPES]
if len(sys.argv) > 2:
    lbm = Lib(sys.argv[2])
    lbm.dll.uh_diag_set.argtypes = [ctypes.c_void_p]; lbm.dll.uh_diag_set.restype = ctypes.c_int
    jobs += [(lbm, sh, False) for sh in MIN_SHAPES]
for lb, (B, H, W, Ci, Co, want_stats), full_stamps in jobs:
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(B, H, W, Ci, generator=g)).to(dev, torch.bfloat16)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(dev)
    wf = torch.empty(Co * 9 * Ci, dtype=torch.bfloat16, device=dev); wd = torch.empty_like(wf)
    lb.call("uh_pack_w3x3", w.data_ptr(), *w.stride(), Co, Ci, wf.data_ptr(), wd.data_ptr(), L.UH_BF16 | 0x300, st)
    y = torch.empty(B, H, W, Co, dtype=torch.bfloat16, device=dev)
    ns = lb.query("uh_conv3x3_stat_slabs", B, H, W, Ci, Co, L.UH_BF16)
    stats = torch.zeros(ns * (2 * Co + 2), dtype=torch.float32, device=dev)
    dbg = torch.zeros(2048 * 4 * 66, dtype=torch.int64, device=dev)
    def run():
        lb.call("uh_conv3x3_fwd", x.data_ptr(), Ci, Ci, None, 0, 0, wf.data_ptr(), y.data_ptr(), Co, Co, stats.data_ptr() if want_stats else None,
                B, H, W, L.UH_BF16 | 0x100, st)
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
    d = dbg.view(-1, 66).cpu().numpy()
    d = d[d[:, 0] > 0]
    n = int(d[0, 0]); assert (d[:, 0] == n).all(), set(d[:, 0].tolist())
    t = d[:, 2:2 + n].astype(np.float64)
    total = t[:, n - 1] - t[:, 0]
    clk = total / (d[:, 1].astype(np.float64) * 10.0)              # core cycles per ns: s_memtime ticks / (100 MHz ticks * 10 ns)
    nchunk = Ci // 32
    tiles_all = B * ((H + 15) // 16) * ((W + 15) // 16) * (Co // 128 if (Co % 128 == 0 and B * ((H + 15) // 16) * ((W + 15) // 16) * (Co // 128) >= 512) else Co // 64)
    nbw = 2 if (Co % 128 == 0 and B * ((H + 15) // 16) * ((W + 15) // 16) * (Co // 128) >= 512) else 1
    mf_all = 16.0 * 16 * nbw * 9 * nchunk * tiles_all * 4 / 1024   # MFMA issue cycles per SIMD, whole launch
    print(f"== {Ci}->{Co} @{H}x{W} B{B} stats={want_stats} [{'phase stamps' if full_stamps else 'entry/exit stamps only'}]: {us:.1f} us/launch, "
          f"{2.0 * B * H * W * Co * 9 * Ci / us / 1e6:.0f} TFLOP/s; {len(t)} waves")
    print(f"   in-kernel clock (median over waves) {np.median(clk):.3f} GHz; wave lifetime cycles mean / p10 / p90: {total.mean():.0f} / {np.percentile(total, 10):.0f} / "
          f"{np.percentile(total, 90):.0f}; MFMA issue per SIMD {mf_all:.0f} cycles = {mf_all / np.percentile(total, 90):.2f} of the p90 lifetime, "
          f"{mf_all / (us * 1e3 * np.median(clk)):.2f} of the launch at that clock")
    if not full_stamps:
        continue
    per_tile = 2 * nchunk + 3
    ntl = (n - 2) // per_tile
    assert 2 + ntl * per_tile == n, (n, per_tile)
    mma = np.zeros(len(t)); fence = np.zeros(len(t)); rnd = np.zeros(len(t)); sto = np.zeros(len(t)); sts = np.zeros(len(t))
    for k in range(ntl):
        o = 2 + k * per_tile
        prev = t[:, o - 1]
        for c in range(nchunk):
            mma += t[:, o + 2 * c] - prev
            fence += t[:, o + 2 * c + 1] - t[:, o + 2 * c]
            prev = t[:, o + 2 * c + 1]
        e = o + 2 * nchunk
        rnd += t[:, e] - prev
        sto += t[:, e + 1] - t[:, e]
        sts += t[:, e + 2] - t[:, e + 1]
    for name, v in (("prologue (entry -> first fence)", t[:, 1] - t[:, 0]), ("column shifts (MFMA + LDS reads)", mma), ("chunk fences (vmcnt(0) + barrier)", fence),
                    ("rounding (before stores)", rnd), ("stores issue", sto), ("statistics", sts)):
        print(f"   {name:36s} {v.mean():8.0f}  ({v.mean() / total.mean() * 100:4.1f} %)   p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}")
