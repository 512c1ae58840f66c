import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import unet_amd
from oracle import unet_ref as U, losses_ref as L
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = unet_amd.UNet(1, 1, bilinear=True)
g = torch.Generator().manual_seed(1)
images = torch.rand(2, 1, 64, 64, generator=g)
masks = torch.randint(0, 3, (2, 64, 64), generator=g)
st = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
keys = U.param_keys(st)
w = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in st.items()}
x1 = U.double_conv(images.double(), w, 'inc', True, {})
x2 = U.down(x1, w, 'down1', True, {}); x3 = U.down(x2, w, 'down2', True, {}); x4 = U.down(x3, w, 'down3', True, {}); x5 = U.down(x4, w, 'down4', True, {})
u1 = U.up(x5, x4, w, 'up1', True, True, {}); u2 = U.up(u1, x3, w, 'up2', True, True, {}); u3 = U.up(u2, x2, w, 'up3', True, True, {})
u4 = U.up(u3, x1, w, 'up4', True, True, {}); u4.retain_grad()
lg = U.out_conv(u4, w, 'outc')
t = (masks // 2).double()
loss = L.bce_with_logits_mean(lg.squeeze(1), t) + L.dice_loss(torch.sigmoid(lg.squeeze(1)), t)
loss.backward()
real_cot = u4.grad.detach().clone()
A, Bsk = u3.detach().clone(), x1.detach().clone()
up4_state = {k[4:]: v.detach() for k, v in st.items() if k.startswith('up4.')}
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm())
def trial(name, a, bsk, cot):
    ww = {('x.' + k): (v.clone().requires_grad_(True) if not ('running' in k or 'num_b' in k) else v) for k, v in up4_state.items()}
    ad, bd = a.clone().requires_grad_(True), bsk.clone().requires_grad_(True)
    y = U.up(ad, bd, ww, 'x', True, True, {})
    y.backward(cot)
    blk = unet_amd.Up(128, 64, True)
    blk.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in up4_state.items()})
    blk = blk.to(dev).train()
    ag, bg = a.float().to(dev).requires_grad_(True), bsk.float().to(dev).requires_grad_(True)
    yg = blk(ag, bg)
    yg.backward(cot.float().to(dev))
    print('%-28s y %.2e dx1 %.2e dx2 %.2e' % (name, rl2(yg, y.detach()), rl2(ag.grad, ad.grad), rl2(bg.grad, bd.grad)),
          ' '.join('%s %.1e' % (k.split('double_conv.')[-1], rl2(p.grad, ww['x.' + k].grad)) for k, p in blk.named_parameters()))
gg = torch.Generator().manual_seed(7)
trial('real in, real cot', A, Bsk, real_cot)
trial('real in, real cot x1e4', A, Bsk, real_cot * 1e4)
trial('real in, randn cot', A, Bsk, torch.randn(real_cot.shape, generator=gg).double())
trial('randn-relu in, real cot', torch.relu(torch.randn(A.shape, generator=gg)).double(), torch.relu(torch.randn(Bsk.shape, generator=gg)).double(), real_cot)
trial('randn-relu in, randn cot', torch.relu(torch.randn(A.shape, generator=gg)).double(), torch.relu(torch.randn(Bsk.shape, generator=gg)).double(), torch.randn(real_cot.shape, generator=gg).double())
