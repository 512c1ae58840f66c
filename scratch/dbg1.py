import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import unet_amd
from oracle import unet_ref as U, losses_ref as L
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, C, H, W = 2, 64, 64, 64
mod = torch.nn.Sequential()
dc = unet_amd.DoubleConv(128, 64)   # like up4.conv: 128 -> 64 (mid 64)
oc = unet_amd.OutConv(64, 1)
x = torch.rand(B, 128, H, W) * 2
masks = torch.randint(0, 3, (B, H, W))
st = {('dc.' + k): v.detach().clone().double() if v.is_floating_point() else v.clone() for k, v in dc.state_dict().items()}
st.update({('oc.' + k): v.detach().clone().double() for k, v in oc.state_dict().items()})
keys = U.param_keys(st)
work = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in st.items()}
xd = x.double().requires_grad_(True)
h = U.double_conv(xd, work, 'dc', True, {})
h.retain_grad()
lg = U.out_conv(h, work, 'oc')
lg.retain_grad()
t = (masks // 2).double()
loss = L.bce_with_logits_mean(lg.squeeze(1), t) + L.dice_loss(torch.sigmoid(lg.squeeze(1)), t)
loss.backward()
# GPU
dc, oc = dc.to(dev), oc.to(dev)
xg = x.to(dev).requires_grad_(True)
hg = dc(xg); hg.retain_grad()
lgg = oc(hg); lgg.retain_grad()
terms = unet_amd.seg_loss(lgg, masks.to(dev), 1, boundary_weight=0.0)
terms['loss'].backward()
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
print('loss', float(terms['loss']), float(loss))
print('logits', rel(lgg, lg.detach()))
print('dlogits', rel(lgg.grad, lg.grad))
print('dh', rel(hg.grad, h.grad))
print('dx', rel(xg.grad, xd.grad))
for k, p in list(dc.named_parameters()) + [('oc.' + k, p) for k, p in oc.named_parameters()]:
    kk = k if k.startswith('oc.') else 'dc.' + k
    print(kk, rel(p.grad, work[kk].grad))
