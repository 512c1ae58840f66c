"""Dice after N train steps of UNet_T(1,1,bilinear) on the bench's synthetic ellipse batches, per initialisation and per step count:
how much of the bench line's dice_vs_ref is the binary and how much is where a chaotic trajectory happens to be at step 200."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from oracle import unet_ref as U
dev = torch.device("cuda:0")
train = [unet_amd.ellipse_batch(4, 64, seed=100 + i) for i in range(4)]
held = unet_amd.ellipse_batch(8, 64, seed=7)
widths = (8, 16, 32, 64, 128)
for amp in (True, False):
    for seed in range(6):
        sd = U.init_state(1, 1, True, widths=widths, seed=seed)
        model = unet_amd.UNet_T(1, 1, bilinear=True)
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        model = model.to(dev)
        st = unet_amd.TrainStepper(model, lr=1e-4, amp=amp)
        row = []
        for i in range(300):
            im, mk = train[i % 4]
            st.step(im.to(dev), mk.to(dev))
            if (i + 1) in (150, 175, 200, 225, 250, 300):
                d, _, _ = unet_amd.evaluate(model, [{"image": held[0], "mask": held[1]}], dev, amp=amp, postprocess=False)
                row.append(round(float(d), 4))
                model.train()
        st.optimizer.close()
        print("bf16" if amp else "fp32", "seed", seed, "dice at 150/175/200/225/250/300 steps:", row, flush=True)
