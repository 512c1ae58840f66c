import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device('cuda:0')
im, mk = unet_amd.ellipse_batch(2, 64, seed=9)
for ctor, bil, amp in [(unet_amd.UNet_T, False, False), (unet_amd.UNet_T, True, False), (unet_amd.UNet, True, True), (unet_amd.UNet, False, True)]:
    res = []
    for rep in range(2):
        torch.manual_seed(0)
        model = ctor(1, 1, bilinear=bil).to(dev)
        st = unet_amd.TrainStepper(model, lr=1e-4, amp=amp)
        t = st.step(im.to(dev), mk.to(dev))
        torch.cuda.synchronize()
        g = st.optimizer.flat_g.clone()
        res.append((float(t["loss"].detach()), g, {k: v.clone() for k, v in model.state_dict().items()}, t["logits"].clone()))
    bad = [k for k in res[0][2] if not torch.equal(res[0][2][k], res[1][2][k])]
    print(ctor.__name__, "bilinear" if bil else "convT", "amp" if amp else "fp32", "loss equal", res[0][0] == res[1][0], "logits equal", torch.equal(res[0][3], res[1][3]),
          "grads equal", torch.equal(res[0][1], res[1][1]), "params differing:", len(bad), bad[:4], flush=True)
