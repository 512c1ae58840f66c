import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd import ops
dev = torch.device('cuda:0')
im, mk = unet_amd.ellipse_batch(2, 64, seed=9)
torch.manual_seed(0)
model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev).train()
x = ops.to_nhwc(im.to(dev), torch.float32)
def run():
    outs = {}
    with torch.no_grad():
        h = model.inc.nhwc(x); outs['inc'] = h.clone()
        skips = []
        for k in range(1, 5):
            s, h = getattr(model, f"down{k}").nhwc_with_skip(h); skips.append(s); outs[f'down{k}'] = h.clone()
        for j in range(1, 5):
            h = getattr(model, f"up{j}").nhwc(h, skips[4 - j]); outs[f'up{j}'] = h.clone()
        outs['out'] = model.outc.nhwc(h).clone()
    return outs
a = run(); b = run(); c = run()
for k in a:
    print(k, tuple(a[k].shape), "ab equal", torch.equal(a[k], b[k]), "bc equal", torch.equal(b[k], c[k]), "max|a-b|", float((a[k].float() - b[k].float()).abs().max()))
# first conv only, repeated
seq = model.inc.double_conv
from unet_amd.unet.unet_parts import _conv_bn_relu
outs = [ _conv_bn_relu(x, None, seq[0], seq[1], True, keep_padded=True).clone() for _ in range(4)]
print("first conv repeat equal:", [torch.equal(outs[0], o) for o in outs[1:]], "pad channels max", float(outs[0][..., 8:].abs().max()))
