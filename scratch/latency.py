import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device('cuda:0')
for name, ctor in [("UNet_T", unet_amd.UNet_T), ("UNet_S", unet_amd.UNet_S), ("UNet", unet_amd.UNet)]:
    for bil in (True, False):
        torch.manual_seed(0)
        m = ctor(1, 3, bilinear=bil).to(memory_format=torch.channels_last).to(dev).eval()
        x = torch.rand(1, 1, 512, 512).to(dev).contiguous(memory_format=torch.channels_last)
        def eager():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                return m(x)
        for _ in range(5): ref = eager()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): eager()
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / 30 * 1e3
        g = unet_amd.GraphedForward(m, x)
        out = g(x)
        torch.cuda.synchronize()
        same = torch.equal(out.float(), ref.float())
        t0 = time.perf_counter()
        for _ in range(30): g(x)
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 30 * 1e3
        print(f"{name} bilinear={bil}: eager {te:.3f} ms, graph {tg:.3f} ms, identical={same}", flush=True)
