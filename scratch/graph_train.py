import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device('cuda:0')
for name, ctor, bil, amp, B, S in [("UNet_T fp32", unet_amd.UNet_T, True, False, 2, 64), ("UNet_S bf16", unet_amd.UNet_S, False, True, 8, 512), ("UNet bf16", unet_amd.UNet, True, True, 8, 512)]:
    im, mk = unet_amd.ellipse_batch(B, S, seed=9)
    im = im.to(dev).contiguous(memory_format=torch.channels_last); mk = mk.to(dev)
    res = []
    for cls in (unet_amd.TrainStepper, unet_amd.GraphedTrainStepper):
        torch.manual_seed(0)
        m = ctor(1, 1, bilinear=bil).to(memory_format=torch.channels_last).to(dev)
        st = cls(m, lr=1e-4, amp=amp)
        for _ in range(3): t = st.step(im, mk)
        torch.cuda.synchronize()
        snap = st.optimizer.flat_p.clone(); loss = float(t["loss"].detach())
        t0 = time.perf_counter()
        for _ in range(20): st.step(im, mk)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        res.append((snap, loss, ms, {k: v.clone() for k, v in m.state_dict().items() if 'running' in k or 'num_batches' in k}))
    same = torch.equal(res[0][0], res[1][0]) and all(torch.equal(res[0][3][k], res[1][3][k]) for k in res[0][3])
    print(f"{name}: eager {res[0][2]:.2f} ms/step, graph {res[1][2]:.2f} ms/step, params+buffers after 3 steps identical: {same}, loss {res[0][1]:.5f} / {res[1][1]:.5f}", flush=True)
