"""The eight backward-data launches of config 2 that also form BatchNorm-backward sums, standalone on seeded data: plain backward-data
(uh_conv3x3_fwd with the backward-data pack) against uh_conv3x3_dgrad_bnsum, us per launch (HIP events, 20 launches after 5).
    [UH_LIB_PATH=variant.so] python scratch/r4_bsum_bench.py [batch]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("unet-medical-image-contour-segmentation_amd")
ops = importlib.import_module("unet-medical-image-contour-segmentation_amd.ops")
LIB = ops.LIB
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# (name, H, channels of dy (the conv's Cout), channels of dx = q (its Cin))
LAYERS = [("down1.3", 256, 128, 128), ("down2.3", 128, 256, 256), ("down3.3", 64, 512, 512), ("down4.3", 32, 512, 512),
          ("up1.3", 64, 256, 512), ("up2.3", 128, 128, 256), ("up3.3", 256, 64, 128), ("up4.3", 512, 64, 64)]
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0]
for name, H, Cdy, Cdx in LAYERS:
    W = H
    dy = (torch.randn(B, H, W, Cdy, device=dev) * 0.1).bfloat16()
    q = torch.randn(B, H, W, Cdx, device=dev).bfloat16()
    w = torch.randn(Cdy, Cdx, 3, 3, device=dev) * 0.05
    dt = ops.UH_BF16
    frag_d = ops.wfrag_ok(B, H, W, Cdy, 0, Cdx, Cdy, 0, Cdx, dt)
    _, wd = ops.pack_w3x3(w, torch.bfloat16, True, frag_d=frag_d)
    coef = torch.cat([torch.rand(Cdx, device=dev) + 0.5, torch.randn(Cdx, device=dev) * 0.1, torch.randn(Cdx, device=dev) * 0.1,
                      torch.rand(Cdx, device=dev) + 0.5]).contiguous()
    dx = torch.empty(B, H, W, Cdx, dtype=torch.bfloat16, device=dev)
    rows = LIB.query("uh_conv3x3_dgrad_bnsum_rows", B, H, W, Cdy, Cdx, Cdy, Cdx, Cdx, dt)
    bsum = torch.empty(max(rows, 1) * 2 * Cdx, dtype=torch.float32, device=dev)
    flag = dt | (ops.UH_WFRAG if frag_d else 0)
    st = ops._stream()

    def plain():
        LIB.call("uh_conv3x3_fwd", dy.data_ptr(), Cdy, Cdy, None, 0, 0, wd.data_ptr(), dx.data_ptr(), Cdx, Cdx, None, B, H, W, flag, st)

    def fused():
        LIB.call("uh_conv3x3_dgrad_bnsum", dy.data_ptr(), Cdy, Cdy, wd.data_ptr(), dx.data_ptr(), Cdx, Cdx, q.data_ptr(), Cdx,
                 coef.data_ptr(), bsum.data_ptr(), B, H, W, flag, st)

    def reduce():
        nblk = LIB.query("uh_bn_bwd_nblk", B * H * W, Cdx)
        part = torch.empty(nblk * 2 * Cdx, dtype=torch.float32, device=dev)
        LIB.call("uh_bn_relu_bwd_reduce", dx.data_ptr(), Cdx, q.data_ptr(), Cdx, coef.data_ptr(), coef[Cdx:].data_ptr(),
                 coef[2 * Cdx:].data_ptr(), coef[3 * Cdx:].data_ptr(), part.data_ptr(), B * H * W, Cdx, dt, st)

    tp, tf = timeit(plain), (timeit(fused) if rows > 0 else float("nan"))
    tr = timeit(reduce)
    gf = 2.0 * B * H * W * Cdx * 9 * Cdy / 1e9
    qmb = B * H * W * Cdx * 2 / 1e6
    tot[0] += tp
    tot[1] += tf
    print(f"{name:8s} H={H:3d} dy {Cdy:3d} -> dx {Cdx:3d}  plain {tp:7.1f} us ({gf / tp * 1e-3 * 1e3:6.0f} TF)  fused {tf:7.1f} us  extra {tf - tp:6.1f} us"
          f"  (q = {qmb:5.0f} MB = {qmb / 5.5e3 * 1e3:5.1f} us at 5.5 TB/s; separate reduce pass {tr:6.1f} us)", flush=True)
print(f"sum plain {tot[0]:.1f} us, fused {tot[1]:.1f} us, extra {tot[1] - tot[0]:.1f} us")
