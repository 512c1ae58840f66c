"""BatchNorm-backward reduce kernels in isolation (TB/s): plain, pool tail, head tail, recomputed stem."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd._lib import LIB
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, nbytes, what):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"  {what}: {us:.1f} us  {nbytes / us / 1e6:.2f} TB/s")


for (B, H, W, C) in [(8, 512, 512, 64), (8, 256, 256, 128), (8, 128, 128, 256)]:
    n = B * H * W
    dz = torch.randn(n, C, device=dev).bfloat16(); y = torch.randn(n, C, device=dev).bfloat16()
    dp = torch.randn(n // 4, C, device=dev).bfloat16()
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev); mu = torch.randn(C, device=dev); rs = torch.rand(C, device=dev) + 0.5
    nblk = LIB.query("uh_bn_bwd_nblk", n, C)
    part = torch.empty(nblk * 2 * C + 64, device=dev)
    timeit(lambda: LIB.call("uh_bn_relu_bwd_reduce", dz.data_ptr(), C, y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), part.data_ptr(), n, C, 1, st),
           2 * n * C * 2, f"plain reduce C={C} {H}x{W}")
    timeit(lambda: LIB.call("uh_bn_relu_pool_bwd_reduce", dz.data_ptr(), C, dp.data_ptr(), C, y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), part.data_ptr(), B, H, W, C, 1, st),
           (2 * n + n // 4) * C * 2, f"pool reduce  C={C} {H}x{W}")
