"""The BatchNorm + ReLU between the two convs of a DoubleConv applied by the CONSUMER conv's loaders (uh_conv3x3_fwd_pre /
uh_conv3x3_wgrad_pre): the activation is never stored, and every result must be BIT-identical
to the path that stores it (uh_bn_relu_apply + uh_conv3x3_fwd / uh_conv3x3_wgrad) -- forward output, BatchNorm statistics
rows, filter gradients, and a whole bf16 train step of the full UNet."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

PRE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unet-medical-image-contour-segmentation_amd",
                       "libunet_hip_pre.so")


@pytest.fixture(autouse=True, scope="module")
def _bind_the_pre_build():
    """The PRE instantiations are a measured net loss and live behind a build flag: the default library
    answers uh_conv3x3_pre_ok with 0 and refuses the two entry points.  __graft_entry__.build() builds the flagged library beside
    the default one (same C ABI, own object directory); for the length of this module the ctypes binding points at it, so that
    the driver's one `pytest -m gpu` session covers the kernel family too.
    The PRE instantiations have no K-split form (KS = 2 adds the two halves of the contraction in another order): the
    stored-activation leg of the flagged library takes the unsplit kernel too (UH_NO_KSPLIT, read once per LIBRARY at its first
    conv dispatch), so that "bit-identical" compares the same summation order."""
    from unet_amd import _lib
    if not os.path.exists(PRE_LIB):
        pytest.fail(f"{PRE_LIB} is missing: __graft_entry__.build() builds it")
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    saved = (_lib.LIB._dll, _lib.LIB_PATH, os.environ.get("UH_NO_KSPLIT"))
    os.environ["UH_NO_KSPLIT"] = "1"
    _lib.LIB_PATH, _lib.LIB._dll = PRE_LIB, None
    _lib.LIB.load()
    assert _lib.LIB.query("uh_conv3x3_pre_ok", 2, 64, 64, 128, 256, 128, 256, _lib.UH_BF16), "libunet_hip_pre.so lacks the PRE kernels"
    yield
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _lib.LIB._dll, _lib.LIB_PATH = saved[0], saved[1]
    if saved[2] is None:
        os.environ.pop("UH_NO_KSPLIT", None)
    else:
        os.environ["UH_NO_KSPLIT"] = saved[2]


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    return torch.device("cuda:0")


# B, H, W, C0 (channels of the raw tensor = K of the conv), Cout
SHAPES = [
    (1, 448, 448, 64, 64),       # register-resident filter instantiation (WRES), more tiles than workgroups, interior + border tiles
    (2, 50, 37, 64, 64),         # WRES with partial tiles on both edges
    (2, 96, 80, 128, 64),        # NBW = 1 streaming filter, 4 K-chunks
    (2, 64, 64, 128, 256),       # NBW = 2
    (1, 176, 160, 256, 256),     # NBW = 2, more tiles than workgroups per slab; backward-weights with 128-row tiles (NWR = 4)
    (3, 16, 16, 512, 512),       # the deepest layer that qualifies (512 coefficients pairs in LDS), one-tile images
    (2, 33, 17, 64, 128),        # odd sizes, NWR = 4
    (1, 250, 333, 128, 128),     # odd sizes, many tiles
]


@pytest.mark.parametrize("B,H,W,C0,Cout", SHAPES)
def test_fused_input_is_bit_identical_to_the_stored_activation(B, H, W, C0, Cout):
    from unet_amd import ops
    from unet_amd._lib import LIB, UH_BF16, UH_WFRAG
    dev = _dev()
    g = torch.Generator().manual_seed(B * 7 + H + C0 + Cout)
    dt = torch.bfloat16
    y_prev = torch.randn(B, H, W, C0, generator=g).to(dev, dt)                       # raw output of the "previous conv"
    scale = (torch.rand(C0, generator=g) + 0.5).to(dev)
    shift = (torch.randn(C0, generator=g) * 0.3).to(dev)
    coef = torch.cat([scale, shift, torch.zeros(2 * C0, device=dev)])
    w = (torch.randn(Cout, C0, 3, 3, generator=g) / (3.0 * C0 ** 0.5)).to(dev)
    dy = torch.randn(B, H, W, Cout, generator=g).to(dev, dt)
    assert LIB.query("uh_conv3x3_pre_ok", B, H, W, C0, Cout, C0, Cout, UH_BF16)
    st = torch.cuda.current_stream().cuda_stream
    # reference path: store the activation, convolve it
    z = torch.empty_like(y_prev)
    LIB.call("uh_bn_relu_apply", y_prev.data_ptr(), C0, scale.data_ptr(), shift.data_ptr(), z.data_ptr(), C0, B * H * W, C0, UH_BF16, st)
    frag = ops.wfrag_ok(B, H, W, C0, 0, Cout, C0, 0, Cout, UH_BF16)
    wf, _ = ops.pack_w3x3(w, dt, False, None, frag, False)
    y_ref, st_ref, nslab = ops.conv3x3_fwd(z, None, wf, Cout, True, None, frag)
    # fused path
    y = torch.empty_like(y_ref)
    stats = torch.empty_like(st_ref)
    LIB.call("uh_conv3x3_fwd_pre", y_prev.data_ptr(), C0, C0, coef.data_ptr(), coef[C0:].data_ptr(), wf.data_ptr(), y.data_ptr(),
             Cout, Cout, stats.data_ptr(), B, H, W, UH_BF16 | (UH_WFRAG if frag else 0), st)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref), f"forward differs in {int((y != y_ref).sum())} of {y.numel()} elements"
    cnt_ref = st_ref[nslab * 2 * Cout:nslab * 2 * Cout + nslab]
    cnt = stats[nslab * 2 * Cout:nslab * 2 * Cout + nslab]
    assert torch.equal(cnt, cnt_ref)
    live = cnt_ref > 0
    assert torch.equal(stats[:nslab * 2 * Cout].view(nslab, 2 * Cout)[live], st_ref[:nslab * 2 * Cout].view(nslab, 2 * Cout)[live])
    # backward-weights
    dw_ref = torch.empty(Cout * 9 * C0, dtype=torch.float32, device=dev)
    ops.conv3x3_wgrad(dy, z, None, dw_ref)
    dw = torch.empty_like(dw_ref)
    ops.conv3x3_wgrad_pre(dy, y_prev, coef, dw)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_ref), f"backward-weights differs: max abs {float((dw - dw_ref).abs().max()):.3e}"
    # and the stored activation is what the formula says (guards the reference leg itself)
    want = torch.relu(torch.addcmul(shift, y_prev.float(), scale)).to(dt)            # fma then ReLU then one rounding
    assert float((z.float() - want.float()).abs().max()) <= 2 ** -8 * float(want.float().abs().max())


def test_shapes_outside_the_fused_path_are_refused():
    from unet_amd._lib import LIB, UH_BF16, UH_F32
    assert not LIB.query("uh_conv3x3_pre_ok", 2, 32, 32, 1024, 512, 1024, 512, UH_BF16)      # > 512 coefficient pairs
    assert not LIB.query("uh_conv3x3_pre_ok", 2, 32, 32, 64, 64, 64, 64, UH_F32)             # bf16 only
    assert not LIB.query("uh_conv3x3_pre_ok", 2, 32, 32, 32, 64, 32, 64, UH_BF16)            # 64-channel slabs in backward-weights
    dev = _dev()
    x = torch.zeros(2, 32, 32, 32, dtype=torch.bfloat16, device=dev)
    c = torch.zeros(128, device=dev)
    y = torch.zeros(2, 32, 32, 64, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(64 * 9 * 32, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="uh_conv3x3_pre_ok"):
        LIB.call("uh_conv3x3_fwd_pre", x.data_ptr(), 32, 32, c.data_ptr(), c[32:].data_ptr(), w.data_ptr(), y.data_ptr(), 64, 64,
                 None, 2, 32, 32, UH_BF16, torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("model_name,size", [("UNet", 128), ("UNet", 80), ("UNetConvT", 64)])
def test_train_step_with_and_without_the_fused_input(model_name, size):
    """Three bf16 steps of the full-width UNet (every DoubleConv of <= 512 mid channels takes the fused path, incl. the pool
    and head tails behind it) against the same steps with the activations stored: logits, loss terms, gradient norm,
    parameters and BatchNorm buffers bit-identical."""
    import unet_amd
    from unet_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    im = torch.rand(2, 1, size, size, generator=g).to(dev)
    mk = torch.randint(0, 3, (2, size, size), generator=g).to(dev)
    out = {}
    calls = {}
    default, default_stem, default_bnsum = ops.FUSE_PRE, ops.STEM_RECOMPUTE, ops.FUSE_BNSUM
    ops.STEM_RECOMPUTE = False       # (the recomputed stem sums its BatchNorm-backward partials in another order: not bit-comparable)
    ops.FUSE_BNSUM = False           # (so do the sums formed inside backward-data, which only the stored-activation leg would take)
    for fuse in (False, True):
        ops.FUSE_PRE = fuse
        try:
            torch.manual_seed(0)
            model = unet_amd.UNet(1, 1, bilinear=(model_name == "UNet")).to(memory_format=torch.channels_last).to(dev)
            st = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
            n0 = _count_apply(ops)
            for _ in range(3):
                t = st.step(im, mk)
            torch.cuda.synchronize()
            calls[fuse] = _count_apply(ops) - n0
            out[fuse] = {"logits": t["logits"].clone(), "loss": float(t["loss"]), "gn": float(t["grad_norm"]),
                         "p": st.optimizer.flat_p.clone(), "g": st.optimizer.flat_g.clone(),
                         "buf": {k: v.clone() for k, v in model.state_dict().items() if "running" in k}}
            st.optimizer.close()
        finally:
            ops.FUSE_PRE = default
            if fuse:
                ops.STEM_RECOMPUTE = default_stem
                ops.FUSE_BNSUM = default_bnsum
    a, b = out[False], out[True]
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"] and a["gn"] == b["gn"]
    assert torch.equal(a["g"], b["g"]) and torch.equal(a["p"], b["p"])
    for k in a["buf"]:
        assert torch.equal(a["buf"][k], b["buf"][k]), k
    assert calls[True] < calls[False], calls           # the fused run really skipped uh_bn_relu_apply launches


_APPLY = [0]


def _count_apply(ops):
    """Number of uh_bn_relu_apply launches so far (the C-ABI call is counted through a wrapper installed once)."""
    from unet_amd._lib import LIB
    if not getattr(LIB, "_apply_counted", False):
        orig = LIB.call

        def counting(name, *a):
            if name == "uh_bn_relu_apply":
                _APPLY[0] += 1
            return orig(name, *a)
        LIB.call = counting
        LIB._apply_counted = True
    return _APPLY[0]
