import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
dev = torch.device('cuda:0')
for name, ctor in [("UNet_T", unet_amd.UNet_T), ("UNet_S", unet_amd.UNet_S)]:
    torch.manual_seed(0)
    m = ctor(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
    st = unet_amd.TrainStepper(m, lr=1e-5, amp=True)
    x = torch.rand(8, 1, 512, 512).to(dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 3, (8, 512, 512)).to(dev)
    for _ in range(3): st.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): st.step(x, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name} B=8 512^2 bf16 train step {ms:.2f} ms = {8/ms*1e3:.0f} img/s", flush=True)
