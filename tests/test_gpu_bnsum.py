"""BatchNorm-backward sums formed inside backward-data (uh_conv3x3_dgrad_bnsum): the gradient tensor
must be BIT-identical to plain backward-data (uh_conv3x3_fwd with the backward-data filter pack), the per-channel sums must agree
with uh_bn_relu_bwd_reduce + uh_bn_bwd_finalize over the same tensors (another summation order: fp32 tolerance against an fp64
oracle of the same formula, unet_parts.py:16-17 differentiated), and a whole bf16 train step of the full UNet must land where
the unfused one does."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    return torch.device("cuda:0")


# B, H, W, Cdy (channels of the incoming gradient = K of the conv), Cdx (channels of the gradient produced = BatchNorm channels)
SHAPES = [
    (1, 448, 448, 64, 64),       # register-resident filter (WRES): more tiles than workgroups, interior + border tiles
    (2, 50, 37, 64, 64),         # WRES, partial tiles on both edges
    (2, 96, 80, 128, 64),        # NBW = 1 streaming filter (168-register budget)
    (2, 64, 64, 256, 128),       # NBW = 1: too few tiles for the 128-channel slabs
    (1, 176, 160, 256, 256),     # NBW = 2 (128 accumulators), more tiles than workgroups per slab
    (2, 250, 131, 64, 128),      # NBW = 2, odd sizes: clamped rows and columns in the border tiles
    (3, 16, 16, 512, 512),       # one-tile images, deep K (K split inside the 8-wave workgroup, KS = 2)
    (4, 32, 32, 512, 512),       # down4.3 at 4 images per GPU: KS = 2 over 16 tiles x 8 slabs
    (8, 128, 128, 256, 256),     # down2 of BASELINE config 2 at its benchmarked extent
]


@pytest.mark.parametrize("B,H,W,Cdy,Cdx", SHAPES)
def test_gradient_bit_identical_and_sums_match(B, H, W, Cdy, Cdx):
    from unet_amd import ops
    from unet_amd._lib import LIB, UH_BF16, UH_WFRAG
    dev = _dev()
    g = torch.Generator().manual_seed(B * 11 + H + Cdy + 3 * Cdx)
    dt = torch.bfloat16
    dy = torch.randn(B, H, W, Cdy, generator=g).to(dev, dt)
    q = (torch.randn(B, H, W, Cdx, generator=g) * 1.5 + 0.25).to(dev, dt)            # raw output of the first conv
    w2 = (torch.randn(Cdy, Cdx, 3, 3, generator=g) / (3.0 * Cdx ** 0.5)).to(dev)     # the SECOND conv's filter [out=Cdy, in=Cdx]
    gamma = (torch.rand(Cdx, generator=g) + 0.5)
    beta = torch.randn(Cdx, generator=g) * 0.3
    qf = q.float().cpu()
    mean = qf.mean(dim=(0, 1, 2))
    rstd = 1.0 / torch.sqrt(qf.var(dim=(0, 1, 2), unbiased=False) + 1e-5)
    scale = gamma * rstd
    shift = beta - mean * scale
    coef = torch.cat([scale, shift, mean, rstd]).to(dev)
    n = B * H * W
    st = torch.cuda.current_stream().cuda_stream
    frag_d = ops.wfrag_ok(B, H, W, Cdy, 0, Cdx, Cdy, 0, Cdx, UH_BF16)
    _, wd = ops.pack_w3x3(w2, dt, True, None, False, frag_d)
    # reference path: plain backward-data, then the reduce pass
    dx_ref, _, _ = ops.conv3x3_fwd(dy, None, wd, Cdx, False, None, frag_d)
    nblk = LIB.query("uh_bn_bwd_nblk", n, Cdx)
    part_ref = torch.empty(nblk * 2 * Cdx, dtype=torch.float32, device=dev)
    LIB.call("uh_bn_relu_bwd_reduce", dx_ref.data_ptr(), Cdx, q.data_ptr(), Cdx, coef.data_ptr(), coef[Cdx:].data_ptr(),
             coef[2 * Cdx:].data_ptr(), coef[3 * Cdx:].data_ptr(), part_ref.data_ptr(), n, Cdx, UH_BF16, st)
    dg_ref, db_ref = torch.empty(Cdx, device=dev), torch.empty(Cdx, device=dev)
    LIB.call("uh_bn_bwd_finalize", part_ref.data_ptr(), nblk, Cdx, dg_ref.data_ptr(), db_ref.data_ptr(), st)
    # fused path
    rows = LIB.query("uh_conv3x3_dgrad_bnsum_rows", B, H, W, Cdy, Cdx, Cdy, Cdx, Cdx, UH_BF16)
    assert rows > 0
    dx = torch.full_like(dx_ref, float("nan"))
    part = torch.full((rows * 2 * Cdx,), float("nan"), dtype=torch.float32, device=dev)
    LIB.call("uh_conv3x3_dgrad_bnsum", dy.data_ptr(), Cdy, Cdy, wd.data_ptr(), dx.data_ptr(), Cdx, Cdx, q.data_ptr(), Cdx,
             coef.data_ptr(), part.data_ptr(), B, H, W, UH_BF16 | (UH_WFRAG if frag_d else 0), st)
    dg, db = torch.empty(Cdx, device=dev), torch.empty(Cdx, device=dev)
    LIB.call("uh_bn_bwd_finalize", part.data_ptr(), rows, Cdx, dg.data_ptr(), db.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref), f"backward-data differs in {int((dx != dx_ref).sum())} of {dx.numel()} elements"
    assert bool(torch.isfinite(part).all()), "a partial row was not written"
    # fp64 oracle of the two sums over the STORED gradient
    d64, q64 = dx_ref.double().cpu(), q.double().cpu()
    on = (q.float().cpu() * scale + shift) > 0            # fp32 fma in the kernels; a tie at exactly 0 is measure-zero here
    g64 = torch.where(on, d64, torch.zeros_like(d64))
    db64 = g64.sum(dim=(0, 1, 2))
    dg64 = (g64 * (q64 - mean.double()) * rstd.double()).sum(dim=(0, 1, 2))
    mag_b = g64.abs().sum(dim=(0, 1, 2)) + 1e-30
    mag_g = (g64 * (q64 - mean.double()) * rstd.double()).abs().sum(dim=(0, 1, 2)) + 1e-30
    for name, got, ref, want, mag in (("dbeta", db, db_ref, db64, mag_b), ("dgamma", dg, dg_ref, dg64, mag_g)):
        err = ((got.double().cpu() - want).abs() / mag).max().item()
        err_ref = ((ref.double().cpu() - want).abs() / mag).max().item()
        # relative to the sum of magnitudes (what fp32 partial sums can deliver); the fused order must be as good as the pass it replaces
        assert err < 2e-6, f"{name}: fused sums off by {err:.2e} of the magnitude sum (separate pass: {err_ref:.2e})"
        assert err_ref < 2e-6, f"{name}: the reference leg itself is off by {err_ref:.2e}"


def test_shapes_outside_the_fused_path_say_so():
    from unet_amd._lib import LIB, UH_BF16, UH_F32
    assert LIB.query("uh_conv3x3_dgrad_bnsum_rows", 2, 32, 32, 64, 64, 64, 64, 64, UH_F32) == 0      # bf16 only
    assert LIB.query("uh_conv3x3_dgrad_bnsum_rows", 2, 32, 32, 64, 48, 64, 48, 48, UH_BF16) == 0     # MFMA-aligned channel counts only
    assert LIB.query("uh_conv3x3_dgrad_bnsum_rows", 2, 32, 32, 64, 64, 64, 64, 128, UH_BF16) == 0    # q must have the gradient's pitch
    dev = _dev()
    z = torch.zeros(2, 32, 32, 64, dtype=torch.bfloat16, device=dev)
    q = torch.zeros(2, 32, 32, 128, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(64 * 9 * 64, dtype=torch.bfloat16, device=dev)
    c = torch.zeros(256, device=dev)
    part = torch.zeros(4096, device=dev)
    with pytest.raises(RuntimeError, match="uh_conv3x3_dgrad_bnsum_rows"):
        LIB.call("uh_conv3x3_dgrad_bnsum", z.data_ptr(), 64, 64, w.data_ptr(), z.data_ptr(), 64, 64, q.data_ptr(), 128, c.data_ptr(),
                 part.data_ptr(), 2, 32, 32, UH_BF16, torch.cuda.current_stream().cuda_stream)


_REDUCE = [0]


def _count_reduce():
    """Number of uh_bn_relu_bwd_reduce launches so far (the C-ABI call is counted through a wrapper installed once)."""
    from unet_amd._lib import LIB
    if not getattr(LIB, "_reduce_counted", False):
        orig = LIB.call

        def counting(name, *a):
            if name == "uh_bn_relu_bwd_reduce":
                _REDUCE[0] += 1
            return orig(name, *a)
        LIB.call = counting
        LIB._reduce_counted = True
    return _REDUCE[0]


@pytest.mark.parametrize("bilinear,size", [(True, 128), (True, 80), (False, 64)])
def test_train_steps_with_and_without_the_fused_sums(bilinear, size):
    """Three bf16 steps of the full-width UNet with the sums formed inside backward-data against the same steps with the separate
    reduce pass.  The forward pass is untouched (first-step logits and loss bit-equal); the two legs sum the same numbers in another
    order, so gradients agree to fp32 summation noise on the BatchNorm parameters and to bf16 noise downstream of them."""
    import unet_amd
    from unet_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    im = torch.rand(2, 1, size, size, generator=g).to(dev)
    mk = torch.randint(0, 3, (2, size, size), generator=g).to(dev)
    out, calls = {}, {}
    default = ops.FUSE_BNSUM
    for fuse in (False, True):
        ops.FUSE_BNSUM = fuse
        try:
            torch.manual_seed(0)
            model = unet_amd.UNet(1, 1, bilinear=bilinear).to(memory_format=torch.channels_last).to(dev)
            st = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
            n0 = _count_reduce()
            first = None
            for _ in range(3):
                t = st.step(im, mk)
                if first is None:
                    first = (t["logits"].clone(), float(t["loss"].detach()), st.optimizer.flat_g.clone())
            torch.cuda.synchronize()
            calls[fuse] = _count_reduce() - n0
            out[fuse] = {"first": first, "logits": t["logits"].clone(), "loss": float(t["loss"].detach()), "gn": float(t["grad_norm"])}
            st.optimizer.close()
        finally:
            ops.FUSE_BNSUM = default
    a, b = out[False], out[True]
    assert torch.equal(a["first"][0], b["first"][0]) and a["first"][1] == b["first"][1]       # forward untouched
    ga, gb = a["first"][2].double(), b["first"][2].double()
    assert bool(torch.isfinite(gb).all())
    rel = float((ga - gb).norm() / ga.norm())
    assert rel < 2e-2, f"first-step gradients differ by {rel:.3e} (L2, all parameters)"
    assert abs(float(ga.norm()) - float(gb.norm())) <= 5e-3 * float(ga.norm())
    # third step: the two trajectories have taken two RMSprop steps from gradients that differ by bf16 noise (measured: 4e-3 on
    # the loss at 128^2) -- a sanity bound, the first-step checks above are the test
    assert abs(a["loss"] - b["loss"]) <= 2e-2 * abs(a["loss"]) and abs(a["gn"] - b["gn"]) <= 1e-1 * abs(a["gn"])
    # (parameters are not compared: RMSprop's first steps are lr * g / (0.1 |g|) = sign steps of 1e-3 on weights of ~1e-2, so the
    # sign of every near-zero gradient element decides 10 % of a weight -- the third-step loss and gradient norm above are the
    # trajectory check, as in the golden trajectories of test_gpu_parity.py)
    # every DoubleConv but the stem's (recomputed output: its own kernels) hands its first BatchNorm's sums to backward-data
    assert calls[False] - calls[True] == 3 * 8, calls


def test_second_consumer_of_the_activation_falls_back_to_the_reduce_pass():
    """The sums that come with backward-data belong to ONE gradient tensor.  When the activation between the two convs has a second
    reader, autograd hands the first layer the SUM of two gradients (another tensor): the link must be ignored and the reduce pass
    must run on what arrived -- same gradients as with the fusion switched off, bit for bit (both legs then run the same kernels)."""
    from unet_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    B, H, W, C = 2, 48, 40, 64
    x = torch.randn(B, H, W, C, generator=g).to(dev, torch.bfloat16)
    side = torch.randn(B, H, W, C, generator=g).to(dev, torch.bfloat16)
    cot = torch.randn(B, H, W, C, generator=g).to(dev, torch.bfloat16)

    def run(fuse, second_reader):
        torch.manual_seed(4)
        c1 = torch.nn.Conv2d(C, C, 3, padding=1, bias=False).to(dev)
        b1 = torch.nn.BatchNorm2d(C).to(dev)
        c2 = torch.nn.Conv2d(C, C, 3, padding=1, bias=False).to(dev)
        b2 = torch.nn.BatchNorm2d(C).to(dev)
        old = ops.FUSE_BNSUM
        ops.FUSE_BNSUM = fuse
        n0 = _count_reduce()
        try:
            link = ops.BnSumLink() if fuse else None
            xin = x.clone().requires_grad_(True)
            z1 = ops.ConvBnReluFn.apply(xin, None, c1.weight, b1.weight, b1.bias, b1.running_mean, b1.running_var, b1.num_batches_tracked,
                                        True, 0.1, 1e-5, ops.TAIL_NONE, None, None, False, None, link, None)
            z2 = ops.ConvBnReluFn.apply(z1, None, c2.weight, b2.weight, b2.bias, b2.running_mean, b2.running_var, b2.num_batches_tracked,
                                        True, 0.1, 1e-5, ops.TAIL_NONE, None, None, False, None, None, link)
            loss = (z2.float() * cot.float()).sum()
            if second_reader:
                loss = loss + (z1.float() * side.float()).sum()
            loss.backward()
            torch.cuda.synchronize()
        finally:
            ops.FUSE_BNSUM = old
        return [t.grad.float().clone() for t in (xin, c1.weight, b1.weight, b1.bias, c2.weight)], _count_reduce() - n0

    ref, n_ref = run(False, True)
    got, n_got = run(True, True)
    assert n_ref == 2 and n_got == 2, (n_ref, n_got)              # both BatchNorm layers ran their own reduce pass
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # ... and without the second reader the first layer's reduce pass is gone
    _, n_fused = run(True, False)
    assert n_fused == 1, n_fused
