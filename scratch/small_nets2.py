import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device('cuda:0')
for name, ctor, bil, amp, ncls in [("UNet_S convT bf16", unet_amd.UNet_S, False, True, 1), ("UNet_S convT fp32", unet_amd.UNet_S, False, False, 1),
                                   ("UNet_T convT bf16 3cls", unet_amd.UNet_T, False, True, 3), ("UNet_S bil fp32", unet_amd.UNet_S, True, False, 1)]:
    torch.manual_seed(0)
    m = ctor(1, ncls, bilinear=bil).to(memory_format=torch.channels_last).to(dev)
    st = unet_amd.TrainStepper(m, lr=1e-5, amp=amp)
    x = torch.rand(8, 1, 512, 512).to(dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 3, (8, 512, 512)).to(dev)
    for _ in range(3): st.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): st.step(x, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name}: B=8 512^2 train step {ms:.2f} ms = {8/ms*1e3:.0f} img/s", flush=True)
