"""N train steps of UNet_S (B=8, 512^2, bf16) for rocprofv3 --kernel-trace --stats."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
name = sys.argv[1] if len(sys.argv) > 1 else "UNet_S"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = getattr(unet_amd, name)(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
x = torch.rand(8, 1, 512, 512).to(dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 3, (8, 512, 512)).to(dev)
st = unet_amd.TrainStepper(m, lr=1e-5, amp=True, wgrad_stream=False)
for _ in range(n):
    st.step(x, y)
torch.cuda.synchronize()
