"""rms error against fp64 of the bf16-pair-slab backward-weights result vs that of the reference-style rounding of the total to bf16."""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
dev = torch.device("cuda:0")
for B, H, W, Cin, Cout in [(2, 128, 128, 128, 256), (2, 256, 256, 64, 64), (4, 32, 32, 512, 512), (1, 64, 64, 256, 512)]:
    g = torch.Generator().manual_seed(B + H + Cin + Cout)
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g)).bfloat16().float()
    dy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float()
    wd = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x.double(), wd, padding=1), [wd], dy.double())
    dwk = torch.empty(Cout * 9 * Cin, dtype=torch.float32, device=dev)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    ops.conv3x3_wgrad(nhwc(dy), nhwc(x), None, dwk)
    ours = dwk.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).double().cpu()
    e_o = float((ours - ref).pow(2).mean().sqrt()); e_r = float((ref.float().bfloat16().double() - ref).pow(2).mean().sqrt())
    print(f"B{B} {H}x{W} {Cin}->{Cout}: rms error ours {e_o:.3e}, total rounded to bf16 {e_r:.3e}, ratio {e_o / e_r:.2f}; max |err| / max |dW| {float((ours - ref).abs().max() / ref.abs().max()):.2e}", flush=True)
