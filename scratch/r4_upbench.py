"""uh_bn_relu_upsample2x_fwd / uh_upsample2x_bwd at the four Up-block shapes of config 2 (B=8 512^2 bf16): us and TB/s of tensor bytes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd._lib import LIB, UH_BF16
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for h, C in [(32, 512), (64, 256), (128, 128), (256, 64)]:
    x = torch.randn(B, h, h, C, device=dev).bfloat16(); sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev)
    y = torch.empty(B, 2 * h, 2 * h, C, device=dev, dtype=torch.bfloat16); dx = torch.empty_like(x)
    def fwd(): LIB.call("uh_bn_relu_upsample2x_fwd", x.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), y.data_ptr(), C, B, h, h, C, 2 * h, 2 * h, 0, 0, UH_BF16, st)
    def bwd(): LIB.call("uh_upsample2x_bwd", y.data_ptr(), C, dx.data_ptr(), C, B, h, h, C, 2 * h, 2 * h, 0, 0, UH_BF16, st)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nbytes = (x.numel() + y.numel()) * 2
        print(f"{name} B={B} {h}x{h}x{C}: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
