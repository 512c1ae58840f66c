"""The CPU oracle (reference restatement, fp32) on the protocol of scratch/dice_chaos3.py: does the REFERENCE recipe dip too?
Test infrastructure only (imports oracle/).   python scratch/dice_chaos_oracle.py [seeds...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from oracle import step_ref as S
from oracle import unet_ref as U
torch.set_num_threads(8)
train = [unet_amd.ellipse_batch(4, 64, seed=100 + i) for i in range(4)]
held = unet_amd.ellipse_batch(8, 64, seed=7)
widths = (8, 16, 32, 64, 128)
for seed in [int(a) for a in sys.argv[1:]] or [0, 4]:
    st = U.init_state(1, 1, True, widths=widths, seed=seed)
    opt = None
    row = []
    for i in range(300):
        im, mk = train[i % 4]
        st, opt, _ = S.train_step(st, opt, im, mk, n_classes=1, bilinear=True, lr=1e-4)
        if (i + 1) in (150, 175, 200, 225, 250, 300):
            d, _ = S.evaluate_dice(st, held[0], held[1], n_classes=1, bilinear=True)
            row.append(round(float(d), 4))
    print("oracle fp32 seed", seed, "dice at 150/175/200/225/250/300 steps:", row, flush=True)
