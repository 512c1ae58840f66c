"""Every 3x3 conv launch shape of config 2's forward and backward-data passes, standalone on seeded data (plain calls, no fused sums):
us per launch and TFLOP/s (HIP events, 20 launches after 5).
    [UH_LIB_PATH=variant.so] python scratch/r4_conv_bench.py [batch]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("unet-medical-image-contour-segmentation_amd")
ops = importlib.import_module("unet-medical-image-contour-segmentation_amd.ops")
LIB = ops.LIB
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LAYERS = [("inc.3", 512, 64, 64), ("down1.0", 256, 64, 128), ("down1.3", 256, 128, 128), ("down2.0", 128, 128, 256), ("down2.3", 128, 256, 256),
          ("down3.0", 64, 256, 512), ("down3.3", 64, 512, 512), ("down4.0", 32, 512, 512), ("down4.3", 32, 512, 512),
          ("up1.0", 64, 1024, 512), ("up1.3", 64, 512, 256), ("up2.0", 128, 512, 256), ("up2.3", 128, 256, 128),
          ("up3.0", 256, 256, 128), ("up3.3", 256, 128, 64), ("up4.0", 512, 128, 64), ("up4.3", 512, 64, 64)]
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


tot = {"fwd": 0.0, "dgrad": 0.0}
for name, H, Cin, Cout in LAYERS:
    W = H
    dt = ops.UH_BF16
    st = ops._stream()
    gf = 2.0 * B * H * W * Cin * 9 * Cout / 1e9
    line = f"{name:8s} H={H:3d} {Cin:4d} -> {Cout:3d}"
    for kind, ci, co in (("fwd", Cin, Cout), ("dgrad", Cout, Cin)):
        x = (torch.randn(B, H, W, ci, device=dev) * 0.5).bfloat16()
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        frag = ops.wfrag_ok(B, H, W, ci, 0, co, ci, 0, co, dt)
        wf, _ = ops.pack_w3x3(w, torch.bfloat16, False, frag_f=frag)
        y = torch.empty(B, H, W, co, dtype=torch.bfloat16, device=dev)
        stats = None
        if kind == "fwd":
            nslab = LIB.query("uh_conv3x3_stat_slabs", B, H, W, ci, co, dt)
            stats = torch.empty(nslab * (2 * co + 2), dtype=torch.float32, device=dev)
        flag = dt | (ops.UH_WFRAG if frag else 0)
        sp = None if stats is None else stats.data_ptr()

        def run():
            LIB.call("uh_conv3x3_fwd", x.data_ptr(), ci, ci, None, 0, 0, wf.data_ptr(), y.data_ptr(), co, co, sp, B, H, W, flag, st)

        t = timeit(run)
        tot[kind] += t
        mb = (B * H * W * (ci + co) * 2) / 1e6
        line += f" | {kind} {t:7.1f} us {gf / t * 1e3:5.0f} TF"
        del x, y, w, wf
    # backward-weights (the MFMA kernel + its slab reduction)
    dyt = (torch.randn(B, H, W, Cout, device=dev) * 0.1).bfloat16()
    xt = (torch.randn(B, H, W, Cin, device=dev) * 0.5).bfloat16()
    dw = torch.empty(Cout, 3, 3, Cin, dtype=torch.float32, device=dev)
    t = timeit(lambda: ops.conv3x3_wgrad(dyt, xt, None, dw))
    tot["wgrad"] = tot.get("wgrad", 0.0) + t
    line += f" | wgrad+reduce {t:7.1f} us {gf / t * 1e3:5.0f} TF"
    del dyt, xt, dw
    print(line, flush=True)
print(f"sum fwd {tot['fwd']:.1f} us, dgrad {tot['dgrad']:.1f} us, wgrad+reduce {tot['wgrad']:.1f} us")
