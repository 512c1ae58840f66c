"""How sensitive is the 'Dice after N steps' figure of bench.dice_vs_ref to last-bit changes?  Runs the bf16 / fp32 HIP legs
for several step counts with the library given by UH_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device("cuda:0")
train = [unet_amd.ellipse_batch(4, 64, seed=100 + i) for i in range(4)]
held = unet_amd.ellipse_batch(8, 64, seed=7)
for amp in (True, False):
    for steps in (150, 200, 250, 300, 400):
        torch.manual_seed(0)
        model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
        st = unet_amd.TrainStepper(model, lr=1e-4, amp=amp)
        for i in range(steps):
            im, mk = train[i % 4]
            st.step(im.to(dev), mk.to(dev))
        d, _, _ = unet_amd.evaluate(model, [{"image": held[0], "mask": held[1]}], dev, amp=amp, postprocess=False)
        print(os.environ.get("UH_LIB_PATH", "main")[-20:], "amp" if amp else "f32", steps, round(float(d), 4), flush=True)
