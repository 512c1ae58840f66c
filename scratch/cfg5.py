import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = unet_amd.UNet(1, 1, bilinear=False).to(memory_format=torch.channels_last).to(dev)
st = unet_amd.TrainStepper(model, amp=False)
g = torch.Generator().manual_seed(1)
x = torch.rand(4, 1, 512, 512, generator=g).to(dev); m = torch.randint(0, 3, (4, 512, 512), generator=g).to(dev)
for _ in range(3): st.step(x, m)
torch.cuda.synchronize()
