"""bf16-pair slabs (SLAB16) round every per-split partial sum of backward-weights; when the partials are
much larger than their sum (a gradient whose sign follows image regions) the error exceeds the reference's single rounding of
the total.  Measures, per variant (env of THIS process: UH_WGRAD_CONTIG / UH_WGRAD_SLAB_F32):
  (a) op level: dy = region-signed, zero-sum per channel (what BatchNorm backward hands down), x = ReLU'd with a large mean;
      rms error against fp64 / rms error of rounding the fp64 total to bf16
  (b) a real bf16 train step of the full UNet on ellipse batches (foreground / background masks) after `--steps` steps:
      the flat gradient is written to a file so that variants can be compared with each other.
usage: python scratch/r5_slab_structured.py TAG [--steps N]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402
from unet_amd import ops  # noqa: E402

tag = sys.argv[1]
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
dev = torch.device("cuda:0")
out = "/tmp/r5slab"
os.makedirs(out, exist_ok=True)


def nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(dev, dtype)


def structured(B, H, W, Cin, Cout, seed, smooth):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    disc = (((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < (0.3 * min(H, W)) ** 2).float()           # foreground = centred disc
    sign = disc - (1 - disc) * disc.mean() / (1 - disc.mean())                                  # sums to zero over the image
    amp = torch.rand(1, Cout, 1, 1, generator=g) + 0.5
    dy = sign[None, None] * amp + smooth * torch.randn(B, Cout, H, W, generator=g)
    dy = dy - dy.mean(dim=(0, 2, 3), keepdim=True)                                              # BatchNorm backward: zero sum per channel
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g) + 1.5)                                # mean 1.5: partials ~ mean * sum(dy over range)
    x, dy = x.bfloat16().float(), dy.bfloat16().float()
    wd = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (dwref,) = torch.autograd.grad(F.conv2d(x.double(), wd, padding=1), [wd], dy.double())
    dwk = torch.empty(Cout * 9 * Cin, dtype=torch.float32, device=dev)
    ops.conv3x3_wgrad(nhwc(dy, torch.bfloat16), nhwc(x, torch.bfloat16), None, dwk)
    ours = dwk.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).double().cpu()
    e_ours = float((ours - dwref).pow(2).mean().sqrt())
    e_ref = float((dwref.float().bfloat16().double() - dwref).pow(2).mean().sqrt())
    return e_ours / e_ref, float((ours - dwref).abs().max() / dwref.abs().max())


print(f"== {tag}: UH_WGRAD_CONTIG={os.environ.get('UH_WGRAD_CONTIG')} UH_WGRAD_SLAB_F32={os.environ.get('UH_WGRAD_SLAB_F32')}")
for shape in [(2, 256, 256, 64, 64), (8, 256, 256, 64, 128), (2, 128, 128, 128, 256), (4, 64, 64, 256, 512), (8, 512, 512, 64, 64)]:
    for smooth in (0.0, 0.3):
        r, m = structured(*shape, seed=sum(shape), smooth=smooth)
        print(f"structured {shape} noise {smooth}: rms err / rms err of bf16(total) = {r:8.3f}   max err / max |dW| = {m:.2e}", flush=True)

# (b) a real step: every variant loads the SAME weights (trained `steps` steps by the first variant that runs) and takes ONE step
# on the same batch; the clipped per-parameter gradients go to /tmp (69 MB per variant: too large for gpurun_out)
state_path = os.path.join(out, "trained_state.pt")
batches = [unet_amd.ellipse_batch(8, 512, seed=100 + i) for i in range(3)]
torch.manual_seed(0)
model = unet_amd.UNet(1, 1, bilinear=True).to(dev)
if not os.path.exists(state_path):
    st = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
    for i in range(steps):
        im, mk = batches[i % 2]
        t = st.step(im.to(dev), mk.to(dev))
    torch.cuda.synchronize()
    print(f"trained {steps} steps, loss {float(t['loss']):.4f}")
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, state_path)
    st.optimizer.close()
for which in ("init", "trained"):
    torch.manual_seed(0)
    model = unet_amd.UNet(1, 1, bilinear=True)
    if which == "trained":
        model.load_state_dict(torch.load(state_path))
    model = model.to(dev)
    st = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
    im, mk = batches[2]
    t = st.step(im.to(dev), mk.to(dev))
    torch.cuda.synchronize()
    g = {k: st.optimizer.grad_of(p).detach().float().cpu().clone() for k, p in model.named_parameters()}
    torch.save(g, os.path.join(out, f"grads_{which}_{tag}.pt"))
    print(f"{which}: loss {float(t['loss']):.6f} grad norm {float(t['grad_norm']):.6e}", flush=True)
    st.optimizer.close()
