"""bench.py --convt failed with 'uh_convt2x2_dgrad_mfma: dy too large' at 8 x 512 x 512 bf16: print the arguments of every call."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("unet-medical-image-contour-segmentation_amd")
sys.modules["unet_amd"] = pkg
import unet_amd
from unet_amd import ops
dev = torch.device("cuda:0")
orig = ops.LIB.call
def call(name, *a):
    if name == "uh_convt2x2_dgrad_mfma":
        print(name, "lddy", a[1], "lddx", a[4], "B h w", a[5:8], "Cin Cout", a[8:10], "Ho Wo", a[10:12], "dt", a[14], flush=True)
    return orig(name, *a)
ops.LIB.call = call
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
model = unet_amd.UNet(1, 1, bilinear=False).to(memory_format=torch.channels_last).to(dev)
st = unet_amd.TrainStepper(model, lr=1e-5, amp=True)
im = torch.rand(B, 1, 512, 512, device=dev)
mk = torch.randint(0, 3, (B, 512, 512), device=dev)
try:
    r = st.step(im, mk)
    torch.cuda.synchronize()
    print("step ok, loss", float(r["loss"]))
except Exception as e:
    print("FAILED:", e)
