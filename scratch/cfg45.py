import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device('cuda:0')
def run(name, model, B, C, S, ncls, amp, steps=3):
    model = model.to(memory_format=torch.channels_last).to(dev)
    st = unet_amd.TrainStepper(model, amp=amp)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, C, S, S, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    m = torch.randint(0, max(ncls, 3) if ncls == 1 else ncls, (B, S, S), generator=g).to(dev)
    for _ in range(2): out = st.step(x, m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): out = st.step(x, m)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name}: {dt*1e3:.1f} ms/step, {B/dt:.1f} img/s, loss {float(out['loss']):.4f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del st, model
    torch.cuda.empty_cache()
torch.manual_seed(0)
run("cfg5 UNet(1,1,convT) fp32 B=4 512^2", unet_amd.UNet(1, 1, bilinear=False), 4, 1, 512, 1, False)
run("cfg5' UNet(1,1,convT) bf16 B=8 512^2", unet_amd.UNet(1, 1, bilinear=False), 8, 1, 512, 1, True)
run("cfg4 depth-5 (64..2048) 3x1024^2 4 classes bf16 B=2", unet_amd.UNetDepth(3, 4, True, widths=(64,128,256,512,1024,2048)), 2, 3, 1024, 4, True)
run("cfg2 fp32 UNet(1,1,bilinear) B=8 512^2", unet_amd.UNet(1, 1, bilinear=True), 8, 1, 512, 1, False)
