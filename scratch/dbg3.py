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
work = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in st.items()}
def oracle(dtype):
    w = {k: (v.detach().to(dtype).requires_grad_(True) if k in keys else v) for k, v in st.items()}
    xs = {}
    x1 = U.double_conv(images.to(dtype), w, 'inc', True, {}); xs['x1'] = x1
    x2 = U.down(x1, w, 'down1', True, {}); xs['x2'] = x2
    x3 = U.down(x2, w, 'down2', True, {}); xs['x3'] = x3
    x4 = U.down(x3, w, 'down3', True, {}); xs['x4'] = x4
    x5 = U.down(x4, w, 'down4', True, {}); xs['x5'] = x5
    u1 = U.up(x5, x4, w, 'up1', True, True, {}); xs['u1'] = u1
    u2 = U.up(u1, x3, w, 'up2', True, True, {}); xs['u2'] = u2
    u3 = U.up(u2, x2, w, 'up3', True, True, {}); xs['u3'] = u3
    u4 = U.up(u3, x1, w, 'up4', True, True, {}); xs['u4'] = u4
    lg = U.out_conv(u4, w, 'outc'); xs['lg'] = lg
    for v in xs.values(): v.retain_grad()
    t = (masks // 2).to(dtype)
    loss = L.bce_with_logits_mean(lg.squeeze(1), t) + L.dice_loss(torch.sigmoid(lg.squeeze(1)), t)
    loss.backward()
    return xs, w
o64, w64 = oracle(torch.float64)
o32, w32 = oracle(torch.float32)
m = model.to(dev).train()
xg = images.to(dev)
gs = {}
x1 = m.inc(xg); gs['x1'] = x1
x2 = m.down1(x1); gs['x2'] = x2
x3 = m.down2(x2); gs['x3'] = x3
x4 = m.down3(x3); gs['x4'] = x4
x5 = m.down4(x4); gs['x5'] = x5
u1 = m.up1(x5, x4); gs['u1'] = u1
u2 = m.up2(u1, x3); gs['u2'] = u2
u3 = m.up3(u2, x2); gs['u3'] = u3
u4 = m.up4(u3, x1); gs['u4'] = u4
lg = m.outc(u4); gs['lg'] = lg
for v in gs.values(): v.retain_grad()
terms = unet_amd.seg_loss(lg, masks.to(dev), 1, boundary_weight=0.0)
terms['loss'].backward()
def rel(a, b): return float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())
print('%-6s %12s %12s %12s %12s' % ('', 'fwd gpu', 'fwd cpu32', 'grad gpu', 'grad cpu32'))
for k in ['x1','x2','x3','x4','x5','u1','u2','u3','u4','lg']:
    print('%-6s %12.3e %12.3e %12.3e %12.3e' % (k, rel(gs[k], o64[k].detach()), rel(o32[k].detach(), o64[k].detach()), rel(gs[k].grad, o64[k].grad), rel(o32[k].grad, o64[k].grad)))
def rl2(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm())
print('mask mismatches / L2 grad errors')
for k in ['x1','x2','x3','x4','x5','u1','u2','u3','u4']:
    mm = int(((gs[k].detach().cpu() > 0) != (o64[k].detach() > 0)).sum())
    mm32 = int(((o32[k].detach() > 0) != (o64[k].detach() > 0)).sum())
    print('%-4s mask mismatch gpu %d cpu32 %d of %d | grad L2 gpu %.3e cpu32 %.3e' % (k, mm, mm32, gs[k].numel(), rl2(gs[k].grad, o64[k].grad), rl2(o32[k].grad, o64[k].grad)))
worst = sorted(((rel(p.grad, w64[k].grad), rel(w32[k].grad, w64[k].grad), k) for k, p in m.named_parameters()), reverse=True)[:8]
for w in worst: print('%.3e (cpu32 %.3e) %s' % w)
print('L2 worst', sorted(((rl2(p.grad, w64[k].grad), k) for k, p in m.named_parameters()), reverse=True)[:5])
