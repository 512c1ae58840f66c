import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import unet_amd
dev = torch.device('cuda:0')
def run():
    torch.manual_seed(0)
    model = unet_amd.UNet(1, 1, bilinear=True).to(dev)
    g = torch.Generator().manual_seed(1)
    images = torch.rand(2, 1, 64, 64, generator=g).to(dev)
    masks = torch.randint(0, 3, (2, 64, 64), generator=g).to(dev)
    model.train()
    out = model(images)
    terms = unet_amd.seg_loss(out, masks, 1)
    terms['loss'].backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in model.named_parameters()}, out.detach().clone()
a, oa = run()
for it in range(3):
    b, ob = run()
    print('logits equal', bool((oa == ob).all()))
    for k in a:
        if not bool((a[k] == b[k]).all()):
            d = float((a[k] - b[k]).abs().max() / a[k].abs().max())
            print('  DIFF', k, f'{d:.3e}')
