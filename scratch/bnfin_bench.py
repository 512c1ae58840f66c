"""uh_bn_finalize latency vs (C, rows used, rows allocated)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd._lib import LIB
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for C, used, alloc in [(64, 768, 8192), (64, 768, 768), (64, 96, 8192), (128, 512, 2048), (256, 512, 512), (512, 128, 128), (512, 32, 32)]:
    stats = torch.randn(alloc * (2 * C + 2), device=dev)
    cnt = stats[alloc * 2 * C: alloc * 2 * C + alloc]
    cnt.zero_(); cnt[:used] = 256.0
    stats[:alloc * 2 * C].view(alloc, 2, C)[:, 1].abs_()
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    coef = torch.empty(4 * C, device=dev)
    def call():
        LIB.call("uh_bn_finalize", stats.data_ptr(), alloc, C, used * 256, g.data_ptr(), b.data_ptr(), None, None, None, 0.1, 1e-5,
                 coef.data_ptr(), coef[C:].data_ptr(), coef[2 * C:].data_ptr(), coef[3 * C:].data_ptr(), None, st)
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): call()
    e1.record(); torch.cuda.synchronize()
    print(f"C={C:4d} rows used {used:4d} of {alloc:5d}: {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us per call (back to back)")
