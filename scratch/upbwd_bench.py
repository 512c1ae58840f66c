"""uh_upsample2x_bwd latency for the four decoder levels of UNet(1,1,bilinear) B=8 512^2 bf16."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
dev = torch.device("cuda:0")
tot = 0.0
for C, h in [(512, 32), (256, 64), (128, 128), (64, 256)]:
    x = torch.randn(8, h, h, C, device=dev).bfloat16().requires_grad_(True)
    u = ops.UpsampleBilinearPadFn.apply(x, 2 * h, 2 * h)
    dy = torch.randn_like(u)
    for _ in range(3): torch.autograd.grad(u, x, dy, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.autograd.grad(u, x, dy, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    print(f"C={C} {h}->{2*h}: {us:.1f} us")
print(f"total {tot:.1f} us")
