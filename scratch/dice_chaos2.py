import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device("cuda:0")
train = [unet_amd.ellipse_batch(4, 64, seed=100 + i) for i in range(4)]
held = unet_amd.ellipse_batch(8, 64, seed=7)
res = []
for seed in range(8):
    torch.manual_seed(seed)
    model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    st = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
    for i in range(250):
        im, mk = train[i % 4]
        st.step(im.to(dev), mk.to(dev))
    d, _, _ = unet_amd.evaluate(model, [{"image": held[0], "mask": held[1]}], dev, amp=True, postprocess=False)
    res.append(round(float(d), 4))
print(os.environ.get("UH_LIB_PATH", "main")[-20:], "bf16 250 steps, seeds 0..7:", res, "mean", round(sum(res) / len(res), 4), flush=True)
