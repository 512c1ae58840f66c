"""Edge-case train steps through the full HIP path (no reference: checks for crashes / NaNs / shape handling)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device('cuda:0')
cases = [
    ("UNet(1,1,bil) B1 512", lambda: unet_amd.UNet(1, 1, True), 1, 1, 512, 512, True),
    ("UNet(1,3,bil) B2 256x320", lambda: unet_amd.UNet(1, 3, True), 2, 1, 256, 320, True),
    ("UNet(3,4,convT) B2 250x333 (odd)", lambda: unet_amd.UNet(3, 4, False), 2, 3, 250, 333, True),
    ("UNet(1,1,convT) fp32 B1 64", lambda: unet_amd.UNet(1, 1, False), 1, 1, 64, 64, False),
    ("UNet_S(1,1,bil) B3 100x100", lambda: unet_amd.UNet_S(1, 1, True), 3, 1, 100, 100, True),
    ("UNet(1,1,bil) B2 16x16 (bottleneck 1x1)", lambda: unet_amd.UNet(1, 1, True), 2, 1, 16, 16, True),
    ("UNetDepth5 (3,4,bil) B1 1024", lambda: unet_amd.UNetDepth(3, 4, True, widths=(64, 128, 256, 512, 1024, 2048)), 1, 3, 1024, 1024, True),
]
for name, ctor, B, C, H, W, amp in cases:
    torch.manual_seed(0)
    m = ctor().to(memory_format=torch.channels_last).to(dev)
    st = unet_amd.TrainStepper(m, lr=1e-5, amp=amp)
    x = torch.rand(B, C, H, W).to(dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 3, (B, H, W)).to(dev)
    for _ in range(2):
        out = st.step(x, y)
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(out["loss"]).all()) and tuple(out["logits"].shape) == (B, m.n_classes, H, W)
    d = unet_amd.evaluate(m, [{"image": x.cpu(), "mask": y.cpu()}], dev, amp=amp, postprocess=(m.n_classes > 1))
    print(f"{name:45s} loss {float(out['loss']):.4f} gnorm {float(out['grad_norm']):.3f} dice {float(d[0]):.4f} {'OK' if ok else 'BAD'}", flush=True)
    assert ok
    del m, st
    torch.cuda.empty_cache()
print("ALL OK")
