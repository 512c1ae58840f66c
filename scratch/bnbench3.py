"""Plain BatchNorm kernels over the layer shapes of the UNet (B = 8): achieved TB/s per kernel and shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd._lib import LIB
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


tot = {"apply": 0.0, "reduce": 0.0, "bwd_apply": 0.0}
for (C, H) in [(64, 512), (128, 256), (64, 256), (256, 128), (128, 128), (512, 64), (256, 64), (512, 32)]:
    B = 8; n = B * H * H
    dz = torch.randn(n, C, device=dev).bfloat16(); y = torch.randn(n, C, device=dev).bfloat16(); z = torch.empty_like(y)
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev); mu = torch.randn(C, device=dev); rs = torch.rand(C, device=dev) + 0.5
    nblk = LIB.query("uh_bn_bwd_nblk", n, C)
    part = torch.empty(nblk * 2 * C + 64, device=dev); dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    nb = n * C * 2
    t1 = timeit(lambda: LIB.call("uh_bn_relu_apply", y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), z.data_ptr(), C, n, C, 1, st))
    t2 = timeit(lambda: LIB.call("uh_bn_relu_bwd_reduce", dz.data_ptr(), C, y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), part.data_ptr(), n, C, 1, st))
    t3 = timeit(lambda: LIB.call("uh_bn_relu_bwd_apply", dz.data_ptr(), C, y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), part.data_ptr(), nblk, dg.data_ptr(), db.data_ptr(), z.data_ptr(), C, n, 0, C, 1, st))
    t4 = timeit(lambda: LIB.call("uh_bn_bwd_finalize", part.data_ptr(), nblk, C, dg.data_ptr(), db.data_ptr(), st))
    print(f"C={C:4d} {H:3d}^2 ({nb / 1e6:6.1f} MB): apply {t1:6.1f} us {2 * nb / t1 / 1e6:5.2f} TB/s | reduce {t2:6.1f} us {2 * nb / t2 / 1e6:5.2f} | bwd_apply+finalize {t3:6.1f} us ({t4:4.1f} finalize) {3 * nb / (t3 - t4) / 1e6:5.2f}")
    tot["apply"] += t1; tot["reduce"] += t2; tot["bwd_apply"] += t3
print(tot)
