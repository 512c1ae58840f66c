import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = unet_amd.UNet(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
st = unet_amd.TrainStepper(m, lr=1e-5, amp=True)
im, mk = unet_amd.ellipse_batch(8, 512, seed=5)
im = im.to(dev).contiguous(memory_format=torch.channels_last); mk = mk.to(dev)
for i in range(400):
    out = st.step(im, mk)
    if i % 50 == 0 or i == 399:
        torch.cuda.synchronize()
        print(i, f"loss {float(out['loss'].detach()):.4f} gnorm {float(out['grad_norm']):.3f} alloc {torch.cuda.memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB", flush=True)
d = unet_amd.evaluate(m, [{"image": im.cpu(), "mask": mk.cpu()}], dev, amp=True)
print("dice on the training batch after 400 steps:", float(d[0]))
