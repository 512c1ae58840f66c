"""UNet_T / UNet_S train-step time with narrow tensors (ops.NARROW_IO) vs the 64-channel zero-padded formulation,
eager and captured in a HIP graph, + kernel-time attribution of one step."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
from unet_amd.train import GraphedTrainStepper
dev = torch.device('cuda:0')
for name, ctor in [("UNet_T", unet_amd.UNet_T), ("UNet_S", unet_amd.UNet_S)]:
    for narrow in (False, True):
        for amp in (True, False):
            ops.NARROW_IO = narrow
            torch.manual_seed(0)
            m = ctor(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
            x = torch.rand(8, 1, 512, 512).to(dev).contiguous(memory_format=torch.channels_last)
            y = torch.randint(0, 3, (8, 512, 512)).to(dev)
            st = unet_amd.TrainStepper(m, lr=1e-5, amp=amp)
            for _ in range(3): st.step(x, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): st.step(x, y)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            line = f"{name} narrow={int(narrow)} {'bf16' if amp else 'fp32'} B=8 512^2 eager {ms:.2f} ms = {8/ms*1e3:.0f} img/s"
            try:
                gs = GraphedTrainStepper(m, lr=1e-5, amp=amp)
                for _ in range(3): gs.step(x, y)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20): gs.step(x, y)
                torch.cuda.synchronize()
                gms = (time.perf_counter() - t0) / 20 * 1e3
                line += f" | graph {gms:.2f} ms = {8/gms*1e3:.0f} img/s"
            except Exception as e:
                line += f" | graph failed: {type(e).__name__}: {str(e)[:120]}"
            print(line, flush=True)
            del st, m
            torch.cuda.empty_cache()
ops.NARROW_IO = True
