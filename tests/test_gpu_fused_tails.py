"""BatchNorm + ReLU fused with its consumer (csrc/bn_fused.hip) against the separate kernels it replaces and against
fp64 stock-PyTorch maths, through the C ABI:

  pool tail: uh_bn_relu_pool_apply        ==  uh_bn_relu_apply + uh_maxpool2_fwd                      (bit for bit)
             uh_bn_relu_pool_bwd_*        ==  uh_maxpool2_bwd + uh_bn_relu_bwd_reduce / _apply        (dy bit for bit)
  head tail: uh_bn_relu_head_fwd          ==  uh_bn_relu_apply + uh_conv1x1_fwd
             uh_bn_relu_head_bwd_*        ==  uh_conv1x1_dgrad / _wgrad + uh_bn_relu_bwd_reduce / _apply (dy bit for bit)
  network  : one train step of the UNet with ops.FUSE_TAILS on and off -- same loss, same gradients.
The reference graph: unet_parts.py:18-20 (BatchNorm, ReLU), :32 (MaxPool2d), :103 (OutConv); unet_model.py:28-38."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    b = b.double().cpu()
    return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _bn_inputs(B, H, W, C, dtype, dev, seed):
    """Random conv output y with a large mean, BatchNorm coefficients from its exact statistics, the ReLU threshold
    inside the data (so that about half of every window is clipped to 0: ties between zeros in every window)."""
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(B, H, W, C, generator=g) * 2 + 5.0
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1 - 2.5 * gamma
    if dtype == torch.bfloat16:
        y = y.bfloat16().float()
    mean = y.double().mean((0, 1, 2))
    var = y.double().var((0, 1, 2), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = (gamma.double() * rstd).float().to(dev)
    shift = (beta.double() - mean * gamma.double() * rstd).float().to(dev)
    return g, y, gamma, beta, y.to(dev, dtype), scale, shift, mean.float().to(dev), rstd.float().to(dev)


POOL_SHAPES = [(2, 32, 32, 64), (1, 64, 48, 128), (3, 6, 10, 256), (2, 2, 2, 512), (8, 128, 128, 64), (1, 18, 34, 8),
               (2, 12, 20, 24)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_skip", [True, False])
@pytest.mark.parametrize("B,H,W,C", POOL_SHAPES)
def test_pool_tail_matches_separate_kernels(dtype, with_skip, B, H, W, C):
    from unet_amd._lib import LIB
    dev = _dev()
    dtc = 1 if dtype == torch.bfloat16 else 0
    if not LIB.query("uh_bn_relu_pool_ok", B, H, W, C, dtc):
        pytest.skip("shape outside the fused kernels (the host falls back to the separate ones)")
    st = torch.cuda.current_stream().cuda_stream
    g, y, gamma, beta, yg, scale, shift, mean, rstd = _bn_inputs(B, H, W, C, dtype, dev, H * W + C)
    n = B * H * W
    # ---- forward
    z_ref, p_ref = torch.empty_like(yg), torch.empty(B, H // 2, W // 2, C, dtype=dtype, device=dev)
    LIB.call("uh_bn_relu_apply", yg.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), z_ref.data_ptr(), C, n, C, dtc, st)
    LIB.call("uh_maxpool2_fwd", z_ref.data_ptr(), C, p_ref.data_ptr(), C, B, H, W, C, dtc, st)
    z, p = torch.full_like(z_ref, 7.0), torch.full_like(p_ref, 7.0)
    LIB.call("uh_bn_relu_pool_apply", yg.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), z.data_ptr(), C, p.data_ptr(), C,
             B, H, W, C, dtc, st)
    assert torch.equal(z, z_ref) and torch.equal(p, p_ref)
    # against stock maths as well (not only against our own kernels)
    zr = F.relu(yg.double() * scale.double() + shift.double())
    assert _rel(z, zr) < (2e-6 if dtype == torch.float32 else 8e-3)
    assert torch.equal(p.float(), F.max_pool2d(z.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))
    # ---- backward
    dpool = torch.randn(B, H // 2, W // 2, C, generator=g).to(dev, dtype)
    dskip = torch.randn(B, H, W, C, generator=g).to(dev, dtype) if with_skip else None
    dz = torch.empty_like(yg)
    LIB.call("uh_maxpool2_bwd", z_ref.data_ptr(), C, dpool.data_ptr(), C, None if dskip is None else dskip.data_ptr(),
             0 if dskip is None else C, dz.data_ptr(), C, B, H, W, C, dtc, st)
    nblk = LIB.query("uh_bn_bwd_nblk", n, C)
    bn = (yg.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr())

    def sums_and_dy(reduce, apply):
        part = torch.empty(nblk * 2 * C, dtype=torch.float32, device=dev)
        dgam, dbet = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        dy = torch.full_like(yg, 3.0)
        reduce(part)
        apply(part, dgam, dbet, dy)
        return dgam, dbet, dy

    ref = sums_and_dy(
        lambda part: LIB.call("uh_bn_relu_bwd_reduce", dz.data_ptr(), C, *bn, part.data_ptr(), n, C, dtc, st),
        lambda part, dg, db, dy: LIB.call("uh_bn_relu_bwd_apply", dz.data_ptr(), C, *bn, part.data_ptr(), nblk, dg.data_ptr(),
                                          db.data_ptr(), dy.data_ptr(), C, n, 0, C, dtc, st))
    sk = (None if dskip is None else dskip.data_ptr(), 0 if dskip is None else C, dpool.data_ptr(), C)
    got = sums_and_dy(
        lambda part: LIB.call("uh_bn_relu_pool_bwd_reduce", *sk, *bn, part.data_ptr(), B, H, W, C, dtc, st),
        lambda part, dg, db, dy: LIB.call("uh_bn_relu_pool_bwd_apply", *sk, *bn, part.data_ptr(), nblk, dg.data_ptr(),
                                          db.data_ptr(), dy.data_ptr(), C, B, H, W, 0, C, dtc, st))
    # per-channel sums: same addends, another order (windows instead of pixel strides)
    big = max(float(ref[0].abs().max()), float(ref[1].abs().max()), 1e-30)
    assert float((got[0] - ref[0]).abs().max()) < 3e-5 * big * 8 and float((got[1] - ref[1]).abs().max()) < 3e-5 * big * 8
    # dy with the SAME sums: bit for bit (dz is rebuilt exactly as uh_maxpool2_bwd stores it)
    dy2 = torch.full_like(yg, 3.0)
    LIB.call("uh_bn_relu_pool_bwd_apply", *sk, *bn, None, 0, ref[0].data_ptr(), ref[1].data_ptr(), dy2.data_ptr(), C, B, H, W, n,
             C, dtc, st)
    assert torch.equal(dy2, ref[2])
    assert _rel(got[2], ref[2].double()) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ncls", [1, 2, 3, 4])
@pytest.mark.parametrize("B,H,W,lanes", [(2, 33, 20, 8), (1, 64, 64, 16), (8, 96, 96, 8), (1, 1, 3, 8)])
def test_head_tail_matches_separate_kernels(dtype, ncls, B, H, W, lanes):
    from unet_amd._lib import LIB
    dev = _dev()
    dtc = 1 if dtype == torch.bfloat16 else 0
    C = lanes * (8 if dtype == torch.bfloat16 else 4)
    assert LIB.query("uh_bn_relu_head_ok", C, ncls, dtc)
    st = torch.cuda.current_stream().cuda_stream
    g, y, gamma, beta, yg, scale, shift, mean, rstd = _bn_inputs(B, H, W, C, dtype, dev, ncls * 100 + H)
    n = B * H * W
    hw = (torch.randn(ncls, C, generator=g) / 8).to(dev)
    hb = torch.randn(ncls, generator=g).to(dev)
    # ---- forward
    z = torch.empty_like(yg)
    LIB.call("uh_bn_relu_apply", yg.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), z.data_ptr(), C, n, C, dtc, st)
    lg_ref = torch.empty(B, H, W, ncls, dtype=torch.float32, device=dev)
    LIB.call("uh_conv1x1_fwd", z.data_ptr(), C, hw.data_ptr(), hb.data_ptr(), lg_ref.data_ptr(), n, C, ncls, dtc, st)
    lg = torch.full_like(lg_ref, 9.0)
    LIB.call("uh_bn_relu_head_fwd", yg.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), hw.data_ptr(), hb.data_ptr(),
             lg.data_ptr(), n, C, ncls, dtc, st)
    assert _rel(lg, lg_ref) < 2e-6
    want = torch.einsum("bhwc,kc->bhwk", z.double(), hw.double()) + hb.double()
    assert _rel(lg, want) < 2e-6
    # ---- backward
    dl = torch.randn(B, H, W, ncls, generator=g).to(dev)
    dz = torch.empty_like(yg)
    LIB.call("uh_conv1x1_dgrad", dl.data_ptr(), hw.data_ptr(), dz.data_ptr(), C, n, C, ncls, dtc, st)
    wsb = LIB.query("uh_conv1x1_wgrad_ws_bytes", n, C, ncls)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dw_ref, db_ref = torch.empty(ncls, C, dtype=torch.float32, device=dev), torch.empty(ncls, dtype=torch.float32, device=dev)
    LIB.call("uh_conv1x1_wgrad", dl.data_ptr(), z.data_ptr(), C, dw_ref.data_ptr(), db_ref.data_ptr(), ws.data_ptr(), wsb, n, C,
             ncls, dtc, st)
    nblk = LIB.query("uh_bn_bwd_nblk", n, C)
    bn = (yg.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr())
    part = torch.empty(nblk * 2 * C, dtype=torch.float32, device=dev)
    dg_ref, dbt_ref = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
    dy_ref = torch.empty_like(yg)
    LIB.call("uh_bn_relu_bwd_reduce", dz.data_ptr(), C, *bn, part.data_ptr(), n, C, dtc, st)
    LIB.call("uh_bn_relu_bwd_apply", dz.data_ptr(), C, *bn, part.data_ptr(), nblk, dg_ref.data_ptr(), dbt_ref.data_ptr(),
             dy_ref.data_ptr(), C, n, 0, C, dtc, st)
    hwsb = LIB.query("uh_bn_relu_head_bwd_ws_bytes", n, C, ncls)
    hws = torch.empty(hwsb, dtype=torch.uint8, device=dev)
    part2 = torch.empty_like(part)
    dw, db = torch.full_like(dw_ref, 5.0), torch.full_like(db_ref, 5.0)
    LIB.call("uh_bn_relu_head_bwd_reduce", dl.data_ptr(), hw.data_ptr(), *bn, part2.data_ptr(), dw.data_ptr(), db.data_ptr(),
             hws.data_ptr(), hwsb, n, C, ncls, dtc, st)
    dg, dbt, dy = torch.empty_like(dg_ref), torch.empty_like(dbt_ref), torch.full_like(dy_ref, 3.0)
    LIB.call("uh_bn_relu_head_bwd_apply", dl.data_ptr(), hw.data_ptr(), *bn, part2.data_ptr(), nblk, dg.data_ptr(), dbt.data_ptr(),
             dy.data_ptr(), C, n, 0, C, ncls, dtc, st)
    assert _rel(dw, dw_ref) < 2e-5 and _rel(db, db_ref) < 2e-5
    assert _rel(dw, torch.einsum("bhwk,bhwc->kc", dl.double(), z.double())) < 2e-5
    big = max(float(dg_ref.abs().max()), float(dbt_ref.abs().max()), 1e-30)
    assert float((dg - dg_ref).abs().max()) < 3e-4 * big and float((dbt - dbt_ref).abs().max()) < 3e-4 * big
    dy2 = torch.full_like(dy_ref, 3.0)
    LIB.call("uh_bn_relu_head_bwd_apply", dl.data_ptr(), hw.data_ptr(), *bn, None, 0, dg_ref.data_ptr(), dbt_ref.data_ptr(),
             dy2.data_ptr(), C, n, n, C, ncls, dtc, st)
    assert torch.equal(dy2, dy_ref)                 # same sums -> bit for bit: dz rebuilt exactly as uh_conv1x1_dgrad stores it
    assert _rel(dy, dy_ref.double()) < (1e-5 if dtype == torch.float32 else 1e-2)
    # a workspace that is too small is refused, not overrun
    with pytest.raises(RuntimeError):
        LIB.call("uh_bn_relu_head_bwd_reduce", dl.data_ptr(), hw.data_ptr(), *bn, part2.data_ptr(), dw.data_ptr(), db.data_ptr(),
                 hws.data_ptr(), 16, n, C, ncls, dtc, st)


def test_shapes_outside_the_fused_kernels_are_refused():
    from unet_amd._lib import LIB
    assert not LIB.query("uh_bn_relu_pool_ok", 2, 33, 32, 64, 1)       # odd height: the last row has no window
    assert not LIB.query("uh_bn_relu_pool_ok", 2, 32, 32, 20, 1)       # channels not a multiple of 16 bytes
    assert not LIB.query("uh_bn_relu_head_ok", 32, 1, 1)
    assert not LIB.query("uh_bn_relu_head_ok", 64, 5, 1)
    assert LIB.query("uh_bn_relu_head_ok", 64, 1, 1) and LIB.query("uh_bn_relu_head_ok", 64, 4, 0)


@pytest.mark.parametrize("amp", [False, True])
@pytest.mark.parametrize("bilinear,n_classes", [(True, 1), (False, 2)])
def test_unet_step_same_with_and_without_fused_tails(amp, bilinear, n_classes):
    """The whole network, one forward + backward: ops.FUSE_TAILS only changes which kernels run."""
    import unet_amd
    from unet_amd import ops
    dev = _dev()
    torch.manual_seed(3)
    net = unet_amd.UNet(3, n_classes, bilinear).to(dev).train()
    x = torch.randn(2, 3, 64, 96, device=dev)
    cot = torch.randn(2, n_classes, 64, 96, device=dev)
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def run(fuse):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        old = ops.FUSE_TAILS
        ops.FUSE_TAILS = fuse
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out = net(x)
            (out.float() * cot).sum().backward()
        finally:
            ops.FUSE_TAILS = old
        return out.detach().float().clone(), {k: p.grad.detach().float().clone() for k, p in net.named_parameters()}, \
            {k: v.detach().float().clone() for k, v in net.state_dict().items() if "running" in k}

    out_f, grads_f, run_f = run(True)
    out_u, grads_u, run_u = run(False)
    assert _rel(out_f, out_u) < 2e-6
    for k in run_u:
        assert torch.equal(run_f[k], run_u[k]), k
    tol = 2e-2 if amp else 2e-4
    for k in grads_u:
        assert _rel(grads_f[k], grads_u[k]) < tol, k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,h,w,C,Ho,Wo", [(2, 32, 32, 512, 64, 64), (1, 33, 17, 64, 67, 35), (3, 5, 7, 128, 10, 14), (8, 64, 64, 256, 128, 128)])
def test_up_tail_is_bit_identical_to_apply_then_upsample(dtype, B, h, w, C, Ho, Wo):
    """uh_bn_relu_upsample2x_fwd (BatchNorm + ReLU + bilinear x2 + F.pad in one pass, the activation never stored) against
    uh_bn_relu_apply followed by uh_upsample2x_fwd: the same roundings in the same order, so every bit must agree -- incl. the
    zero border of an odd skip extent (unet_parts.py:85-88)."""
    from unet_amd import ops
    from unet_amd._lib import LIB, UH_BF16, UH_F32
    dev = _dev()
    g = torch.Generator().manual_seed(B + h * 3 + C)
    dt = UH_BF16 if dtype == torch.bfloat16 else UH_F32
    y = torch.randn(B, h, w, C, generator=g).to(dev, dtype)
    scale = (torch.rand(C, generator=g) + 0.5).to(dev)
    shift = (torch.randn(C, generator=g) * 0.3).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    assert LIB.query("uh_bn_relu_upsample2x_ok", B, h, w, C, Ho, Wo, dt)
    pt, pl = ops._pad_geometry(h, w, Ho, Wo)
    z = torch.empty_like(y)
    LIB.call("uh_bn_relu_apply", y.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), z.data_ptr(), C, B * h * w, C, dt, st)
    ref = torch.full((B, Ho, Wo, C), float("nan"), dtype=dtype, device=dev)
    LIB.call("uh_upsample2x_fwd", z.data_ptr(), C, ref.data_ptr(), C, B, h, w, C, Ho, Wo, pt, pl, dt, st)
    got = torch.full_like(ref, float("nan"))
    LIB.call("uh_bn_relu_upsample2x_fwd", y.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), got.data_ptr(), C, B, h, w, C, Ho, Wo,
             pt, pl, dt, st)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got).all())
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {got.numel()} elements differ"
    # and the reference leg is what torch computes from the stored activation (guards the leg itself)
    want = torch.nn.functional.interpolate(z.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    want = torch.nn.functional.pad(want, [pl, Wo - 2 * w - pl, pt, Ho - 2 * h - pt]).permute(0, 2, 3, 1)
    assert float((ref.float() - want).abs().max()) <= (2 ** -7 if dtype == torch.bfloat16 else 1e-5) * max(1.0, float(want.abs().max()))


def test_up_tail_refuses_what_it_cannot_do():
    from unet_amd._lib import LIB, UH_BF16
    assert not LIB.query("uh_bn_relu_upsample2x_ok", 2, 8, 8, 12, 16, 16, UH_BF16)       # channels not a multiple of a 16-byte piece
    assert not LIB.query("uh_bn_relu_upsample2x_ok", 2, 8, 8, 64, 15, 16, UH_BF16)       # target smaller than the up-sampled image
    dev = _dev()
    y = torch.zeros(2, 8, 8, 12, dtype=torch.bfloat16, device=dev)
    c = torch.zeros(12, device=dev)
    out = torch.zeros(2, 16, 16, 12, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="uh_bn_relu_upsample2x_ok"):
        LIB.call("uh_bn_relu_upsample2x_fwd", y.data_ptr(), 12, c.data_ptr(), c.data_ptr(), out.data_ptr(), 12, 2, 8, 8, 12, 16, 16, 0, 0,
                 UH_BF16, torch.cuda.current_stream().cuda_stream)
