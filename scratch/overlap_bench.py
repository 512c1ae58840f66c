"""Do an MFMA-bound conv kernel and an HBM-bound BatchNorm kernel overlap when launched on two streams?
Each alone (N launches), then both together; times in ms for the whole batch of launches."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
from unet_amd._lib import LIB, UH_BF16
dev = torch.device("cuda:0")
B, H, C = 8, 128, 256
x = torch.relu(torch.randn(B, H, H, C, device=dev)).bfloat16()
dy = torch.randn(B, H, H, C, device=dev).bfloat16()
w = torch.randn(C, C, 3, 3, device=dev) / 48
wf, wd = ops.pack_w3x3(w, torch.bfloat16, True)
dwk = torch.empty(C * 9 * C, dtype=torch.float32, device=dev)
nb = LIB.query("uh_conv3x3_wgrad_ws_bytes", B, H, H, C, C, UH_BF16)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
yout = torch.empty_like(x)
big = torch.randn(8, 512, 512, 64, device=dev).bfloat16()
zbig = torch.empty_like(big)
sc = torch.ones(64, device=dev); sh = torch.zeros(64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 40
def conv_w(st):
    LIB.call("uh_conv3x3_wgrad", dy.data_ptr(), C, x.data_ptr(), C, C, None, 0, 0, dwk.data_ptr(), C, ws.data_ptr(), nb, B, H, H, UH_BF16, st.cuda_stream)
def conv_f(st):
    LIB.call("uh_conv3x3_fwd", x.data_ptr(), C, C, None, 0, 0, wf.data_ptr(), yout.data_ptr(), C, C, None, B, H, H, UH_BF16, st.cuda_stream)
def bn(st):
    LIB.call("uh_bn_relu_apply", big.data_ptr(), 64, sc.data_ptr(), sh.data_ptr(), zbig.data_ptr(), 64, 8 * 512 * 512, 64, UH_BF16, st.cuda_stream)
def run(fa, fb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        if fa: fa(s1)
        if fb: fb(s2)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for name, f in [("wgrad 256x256@128^2", conv_w), ("fwd 256->256@128^2", conv_f)]:
    for _ in range(2): run(f, bn)
    a, b, ab = run(f, None), run(None, bn), run(f, bn)
    print(f"{name}: alone {a:.2f} ms, bn_relu_apply(268 MB) alone {b:.2f} ms, together {ab:.2f} ms (sum {a + b:.2f}, max {max(a, b):.2f})")
