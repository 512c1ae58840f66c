"""Phase timeline of conv3x3_fwd_mfma_v2 workgroups (trace build, see trace_patch.py): 64->64 @512^2 B=8 bf16 by default."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
from unet_amd._lib import LIB
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = torch.device("cuda:0")
x = torch.randn(8, S, S, Cin, device=dev).bfloat16()
w = (torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5))
wf, _ = ops.pack_w3x3(w, torch.bfloat16, False)
for _ in range(3):
    y, st, ns = ops.conv3x3_fwd(x, None, wf, Cout, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv3x3_fwd(x, None, wf, Cout, True)
e1.record(); torch.cuda.synchronize()
print(f"kernel {e0.elapsed_time(e1) / 10 * 1e3:.1f} us (with stamps)")
buf = (ctypes.c_ulonglong * (16 * 128))()
dll = LIB.load()
dll.uh_trace_read.argtypes = [ctypes.c_void_p]
rc = dll.uh_trace_read(buf)
assert rc == 0, rc
names = {1: "tile", 2: "dma-issued", 3: "mfma-issued", 4: "barrier", 5: "rounded", 6: "stored", 7: "stats"}
tot = {}
for wg in range(16):
    ev = [(buf[wg * 128 + i] >> 56, buf[wg * 128 + i] & ((1 << 56) - 1)) for i in range(128)]
    ev = [(s, t) for s, t in ev if s]
    if not ev: continue
    line = []
    for (s0, t0), (s1, t1) in zip(ev, ev[1:]):
        d = (t1 - t0) * 0.01          # 100 MHz -> us
        key = f"{names[s0]}->{names[s1]}"
        tot.setdefault(key, []).append(d)
        line.append(f"{names[s1]}+{d:.2f}")
    if wg < 3:
        print(f"WG {wg * 53}: " + " ".join(line[:44]))
print()
for k, v in tot.items():
    print(f"{k:28s} n {len(v):4d}  mean {sum(v)/len(v):6.2f} us  min {min(v):6.2f}  max {max(v):6.2f}")
