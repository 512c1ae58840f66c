"""In-kernel clock probe (UH_ABL_CLK builds): forward conv of the 64->64 512^2 layer, back to back; prints the core clock
workgroup 0 saw (s_memtime ticks / s_memrealtime 100 MHz ticks) and the launch time."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratch.ab_conv import Lib
from unet_amd import _lib as L
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B, H, W, C = 8, 512, 512, 64
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(B, H, W, C, generator=g)).to(dev, torch.bfloat16)
w = (torch.randn(C, C, 3, 3, generator=g) / 24).to(dev)
for arg in sys.argv[1:]:
    name, path = arg.split("=")
    lb = Lib(path)
    wf = torch.empty(C * 9 * C, dtype=torch.bfloat16, device=dev); wd = torch.empty_like(wf)
    lb.call("uh_pack_w3x3", w.data_ptr(), *w.stride(), C, C, wf.data_ptr(), wd.data_ptr(), L.UH_BF16 | 0x300, st)
    y = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dev)
    ns = lb.query("uh_conv3x3_stat_slabs", B, H, W, C, C, L.UH_BF16)
    stats = torch.zeros(ns * (2 * C + 2), dtype=torch.float32, device=dev)
    def run():
        lb.call("uh_conv3x3_fwd", x.data_ptr(), C, C, None, 0, 0, wf.data_ptr(), y.data_ptr(), C, C, stats.data_ptr(), B, H, W, L.UH_BF16 | 0x100, st)
    for _ in range(30): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    nt = B * (H // 16) * (W // 16)
    t = stats[nt * (2 * C + 1):nt * (2 * C + 1) + 512 * 4].view(512, 4).cpu().double()
    s0, s1, cyc, xcc = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    base = s0.min()
    import numpy as np
    st_, en_ = ((s0 - base) * 0.01).numpy(), ((s1 - base) * 0.01).numpy()     # us
    q = lambda a: " ".join(f"{v:6.1f}" for v in np.percentile(a, [0, 10, 50, 90, 100]))
    print(f"{name:10s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us/launch | WG start [min p10 p50 p90 max] {q(st_)} | end {q(en_)} | life {q(en_ - st_)} us | clock {float((cyc / ((s1 - s0) * 10)).median()):.3f} GHz")
    print("   first 16 starts:", " ".join(f"{v:.1f}" for v in st_[:16]), "| xcc", " ".join(str(int(v)) for v in xcc[:16].tolist()))
    hw = xcc.numpy().astype(np.int64)
    xc, hwid = hw >> 16, hw & 0xFFFF
    cu, sh, se = (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
    life = en_ - st_
    print("   life by XCC:", " ".join(f"{x}:{life[xc == x].mean():.1f}({(xc == x).sum()})" for x in range(8)))
    print("   life by SE :", " ".join(f"{x}:{life[se == x].mean():.1f}({(se == x).sum()})" for x in sorted(set(se.tolist()))))
    key = xc * 1000 + se * 100 + sh * 16 + cu
    import collections
    cnt = collections.Counter(key.tolist())
    print("   WGs per CU histogram:", collections.Counter(cnt.values()), "distinct CUs", len(cnt))
    per = {k: life[key == k].mean() for k in cnt}
    one = [per[k] for k in cnt if cnt[k] == 1]; two = [per[k] for k in cnt if cnt[k] == 2]; three = [per[k] for k in cnt if cnt[k] >= 3]
    print(f"   mean life on CUs with 1 WG: {np.mean(one) if one else float('nan'):.1f}  2 WGs: {np.mean(two) if two else float('nan'):.1f}  3+: {np.mean(three) if three else float('nan'):.1f}")
    print("   life vs wg index (mean of 64): ", " ".join(f"{life[i:i + 64].mean():.0f}" for i in range(0, 512, 64)))
