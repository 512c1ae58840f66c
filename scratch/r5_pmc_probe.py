"""Four forward conv shapes of config 2, 3 launches each after a warm-up (for a rocprofv3 --pmc pass): the streaming forms where the
filter fragments cost most (up1.0 1024->512 @64, down2.3 256->256 @128, up4.0 128->64 @512) and the register-resident form (inc.3 64->64 @512).
UH_LIB_PATH selects the library (the shipped one or a diagnostic build of scratch/libs/)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("unet-medical-image-contour-segmentation_amd.ops")
LIB = ops.LIB
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = 8
for name, H, ci, co in [("up1.0", 64, 1024, 512), ("down2.3", 128, 256, 256), ("up4.0", 512, 128, 64), ("inc.3", 512, 64, 64)]:
    dt = ops.UH_BF16
    st = ops._stream()
    x = (torch.randn(B, H, H, ci, device=dev) * 0.5).bfloat16()
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    frag = ops.wfrag_ok(B, H, H, ci, 0, co, ci, 0, co, dt)
    wf, _ = ops.pack_w3x3(w, torch.bfloat16, False, frag_f=frag)
    y = torch.empty(B, H, H, co, dtype=torch.bfloat16, device=dev)
    nslab = LIB.query("uh_conv3x3_stat_slabs", B, H, H, ci, co, dt)
    stats = torch.empty(nslab * (2 * co + 2), dtype=torch.float32, device=dev)
    for _ in range(4):
        LIB.call("uh_conv3x3_fwd", x.data_ptr(), ci, ci, None, 0, 0, wf.data_ptr(), y.data_ptr(), co, co, stats.data_ptr(), B, H, H, dt | (ops.UH_WFRAG if frag else 0), st)
    torch.cuda.synchronize()
    del x, y, w, wf
