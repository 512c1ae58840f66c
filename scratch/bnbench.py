import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
from unet_amd._lib import LIB
dev = torch.device('cuda:0')
tot = 0.0
for (B, H, W, C) in [(8, 512, 512, 64), (8, 256, 256, 128), (8, 128, 128, 256), (8, 64, 64, 512)]:
    n = B * H * W
    dz = torch.randn(n, C, device=dev).bfloat16(); y = torch.randn(n, C, device=dev).bfloat16()
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev); mu = torch.randn(C, device=dev); rs = torch.rand(C, device=dev) + 0.5
    nblk = LIB.query("uh_bn_bwd_nblk", n, C)
    part = torch.empty(nblk * 2 * C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        LIB.call("uh_bn_relu_bwd_reduce", dz.data_ptr(), C, y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), part.data_ptr(), n, C, 1, st)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    print(f"  C={C} {H}x{W}: {us:.1f} us  {2*n*C*2/us/1e6:.2f} TB/s nblk={nblk}")
print(" total", round(tot, 1))
