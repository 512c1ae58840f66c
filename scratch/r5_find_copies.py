"""Where do the ~12 __amd_rocclr_copyBuffer launches (and the one torch bfloat16 copy kernel) per train step come from?
torch.profiler over three steps of config 2, memcpy / copy-kernel events with the Python stack that issued them."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = unet_amd.UNet(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
st = unet_amd.TrainStepper(model, amp=True)
g = torch.Generator().manual_seed(1)
im = torch.rand(8, 1, 512, 512, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
mk = torch.randint(0, 3, (8, 512, 512), generator=g).to(dev)
for _ in range(3):
    st.step(im, mk)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        st.step(im, mk)
    torch.cuda.synchronize()
ev = prof.events()
names = {}
for e in ev:
    n = e.name
    if ("memcpy" in n.lower() or "copy" in n.lower() or "fill" in n.lower() or "cat" in n.lower()) and e.device_type == torch.autograd.DeviceType.CPU:
        stack = [s for s in (e.stack or []) if "unet" in s or "train.py" in s or "ops.py" in s][:3]
        key = (n, str(e.input_shapes)[:60], " <- ".join(stack))
        names[key] = names.get(key, 0) + 1
for k, v in sorted(names.items(), key=lambda kv: -kv[1]):
    print(v, k)
print("---- device-side kernels named copy / fill")
dk = {}
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA and ("copy" in e.name.lower() or "fill" in e.name.lower() or "Memcpy" in e.name or "Memset" in e.name):
        dk[e.name[:80]] = dk.get(e.name[:80], 0) + 1
for k, v in dk.items():
    print(v, k)
