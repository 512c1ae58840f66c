"""N eval-mode forwards of UNet(1,1,bilinear) B=8 512^2 bf16 for rocprofv3 --kernel-trace --stats."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = unet_amd.UNet(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev).eval()
x = torch.rand(8, 1, 512, 512).to(dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(n):
        y = m(x)
torch.cuda.synchronize()
