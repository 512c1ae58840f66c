import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
from unet_amd._lib import LIB
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, nbytes, label):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"  {label:34s} {us:8.1f} us  {nbytes/us/1e6:5.2f} TB/s")
    return us
tot = {}
for (B, H, W, C) in [(8, 512, 512, 64), (8, 256, 256, 128), (8, 128, 128, 256)]:
    n = B * H * W
    print(f"C={C} {H}x{W}")
    a = torch.randn(n, C, device=dev).bfloat16(); b = torch.randn(n, C, device=dev).bfloat16(); o = torch.empty_like(a)
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev); mu = torch.randn(C, device=dev); rs = torch.rand(C, device=dev) + 0.5
    dg = torch.randn(C, device=dev); db = torch.randn(C, device=dev)
    eb = a.element_size()
    tot.setdefault('apply', 0); tot['apply'] += timeit(lambda: LIB.call("uh_bn_relu_apply", a.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), o.data_ptr(), C, n, C, 1, st), 2 * n * C * eb, "bn_relu_apply")
    nblk = LIB.query("uh_bn_bwd_nblk", n, C)
    part = torch.empty(nblk * 2 * C, device=dev)
    LIB.call("uh_bn_relu_bwd_reduce", a.data_ptr(), C, b.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), part.data_ptr(), n, C, 1, st)
    tot.setdefault('bwd_apply', 0); tot['bwd_apply'] += timeit(lambda: LIB.call("uh_bn_relu_bwd_apply", a.data_ptr(), C, b.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), part.data_ptr(), nblk, dg.data_ptr(), db.data_ptr(), o.data_ptr(), C, n, 0, C, 1, st), 3 * n * C * eb, "bn_relu_bwd_apply (+finalize)")
    x4 = a.view(B, H, W, C)
    p = torch.empty(B, H // 2, W // 2, C, device=dev, dtype=a.dtype)
    tot.setdefault('pool_fwd', 0); tot['pool_fwd'] += timeit(lambda: LIB.call("uh_maxpool2_fwd", x4.data_ptr(), C, p.data_ptr(), C, B, H, W, C, 1, st), 1.25 * n * C * eb, "maxpool2_fwd")
    dyp = torch.randn_like(p)
    tot.setdefault('pool_bwd', 0); tot['pool_bwd'] += timeit(lambda: LIB.call("uh_maxpool2_bwd", x4.data_ptr(), C, dyp.data_ptr(), C, b.data_ptr(), C, o.data_ptr(), C, B, H, W, C, 1, st), 3.25 * n * C * eb, "maxpool2_bwd (+skip add)")
    xs = p
    tot.setdefault('up_fwd', 0); tot['up_fwd'] += timeit(lambda: LIB.call("uh_upsample2x_fwd", xs.data_ptr(), C, o.data_ptr(), C, B, H // 2, W // 2, C, H, W, 0, 0, 1, st), 1.25 * n * C * eb, "upsample2x_fwd")
    tot.setdefault('up_bwd', 0); tot['up_bwd'] += timeit(lambda: LIB.call("uh_upsample2x_bwd", a.data_ptr(), C, p.data_ptr(), C, B, H // 2, W // 2, C, H, W, 0, 0, 1, st), 1.25 * n * C * eb, "upsample2x_bwd")
print(tot)
