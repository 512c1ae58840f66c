"""uh_conv1x1_wgrad / uh_conv1x1_fwd latency on the UNet head shape (B=8, 512^2, 64 channels, 1 class, bf16)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd._lib import LIB, UH_BF16
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
npix, Cin, ncls = 8 * 512 * 512, 64, 1
x = torch.relu(torch.randn(npix, Cin, device=dev)).bfloat16()
dl = torch.randn(npix, ncls, device=dev)
w = torch.randn(ncls, Cin, device=dev); bias = torch.zeros(ncls, device=dev)
dw = torch.empty(ncls, Cin, device=dev); db = torch.empty(ncls, device=dev)
logits = torch.empty(npix, ncls, device=dev)
nb = LIB.query("uh_conv1x1_wgrad_ws_bytes", npix, Cin, ncls)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
junk = torch.empty(300 << 20, dtype=torch.uint8, device=dev)     # evicts the caches between calls
def t(fn, n=10):
    tot = 0.0
    for _ in range(n):
        junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
print("wgrad %.1f us" % t(lambda: LIB.call("uh_conv1x1_wgrad", dl.data_ptr(), x.data_ptr(), Cin, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, npix, Cin, ncls, UH_BF16, st)))
print("fwd   %.1f us" % t(lambda: LIB.call("uh_conv1x1_fwd", x.data_ptr(), Cin, w.data_ptr(), bias.data_ptr(), logits.data_ptr(), npix, Cin, ncls, UH_BF16, st)))
print("dgrad %.1f us" % t(lambda: LIB.call("uh_conv1x1_dgrad", dl.data_ptr(), w.data_ptr(), x.data_ptr(), Cin, npix, Cin, ncls, UH_BF16, st)))
