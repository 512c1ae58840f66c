"""Maximum sizes: activations of 2 GiB (= 2^31 bytes, one byte past what a buffer descriptor / 32-bit byte offset can
address).  The LDS-DMA conv kernels hand such tensors to the register-staged kernels with 64-bit addressing; every
elementwise / pooling / resampling kernel walks them with 64-bit pixel offsets.  The checks are size-independent
properties: agreement with the same op on sub-windows that DO take the DMA path, linearity of backward-weights over row
bands, closed-form samples of the resampling ops, adjoint identities, and stock PyTorch on row bands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

H = W = 4096
C = 64          # 4096 * 4096 * 64 bf16 = 2^31 bytes


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def big():
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(1, H, W, C, device=dev, dtype=torch.bfloat16, generator=g)
    assert x.numel() * x.element_size() == 1 << 31
    yield x
    del x
    torch.cuda.empty_cache()


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_conv_forward_and_statistics_at_2gib(big):
    from unet_amd import ops
    dev = big.device
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(C, C, 3, 3, generator=g) / 24.0).to(dev)
    wf, _ = ops.pack_w3x3(w, torch.bfloat16, False)
    y, stats, nslab = ops.conv3x3_fwd(big, None, wf, C, True)
    torch.cuda.synchronize()
    S = 96
    for r0, c0 in [(0, 0), (H - S, W - S), (2040, 1000), (0, W - S), (H - S, 0), (2047, 2047)]:
        ra, rb, ca, cb = max(r0 - 1, 0), min(r0 + S + 1, H), max(c0 - 1, 0), min(c0 + S + 1, W)
        crop = big[:, ra:rb, ca:cb].contiguous()                      # small: takes the LDS-DMA kernel
        yc, _, _ = ops.conv3x3_fwd(crop, None, wf, C, False)
        got = y[:, r0:r0 + S, c0:c0 + S]
        ref = yc[:, r0 - ra:r0 - ra + S, c0 - ca:c0 - ca + S]
        assert _rel(got, ref) < 1e-2, (r0, c0, _rel(got, ref))
    # BatchNorm statistics of the stored output: pixel counts, mean, M2 against chunked fp64 sums
    st = stats[:nslab * 2 * C].view(nslab, 2, C).double()
    cnt = stats[nslab * 2 * C:nslab * 2 * C + nslab].double()
    assert float(cnt.sum()) == H * W
    live = cnt > 0
    st, cnt = st[live], cnt[live]
    mean = (st[:, 0] * cnt[:, None]).sum(0) / cnt.sum()
    m2 = (st[:, 1] + cnt[:, None] * (st[:, 0] - mean[None]) ** 2).sum(0)
    s1 = torch.zeros(C, dtype=torch.float64, device=dev)
    s2 = torch.zeros(C, dtype=torch.float64, device=dev)
    for r in range(0, H, 256):
        blk = y[0, r:r + 256].double().reshape(-1, C)
        s1 += blk.sum(0)
        s2 += (blk * blk).sum(0)
    mref = s1 / (H * W)
    m2ref = s2 - s1 * mref
    assert float((mean - mref).abs().max()) < 1e-5
    assert float(((m2 - m2ref).abs() / m2ref).max()) < 1e-4


def test_conv_backward_weights_at_2gib_is_the_sum_over_row_bands(big):
    from unet_amd import ops
    dev = big.device
    g = torch.Generator(device=dev).manual_seed(6)
    dy = torch.randn(1, H, W, C, device=dev, dtype=torch.bfloat16, generator=g)
    full = torch.empty(C * 9 * C, dtype=torch.float32, device=dev)
    ops.conv3x3_wgrad(dy, big, None, full)
    acc = torch.zeros_like(full)
    nb = 4
    rows = H // nb
    for k in range(nb):
        a, b = k * rows, (k + 1) * rows
        ra, rb = max(a - 1, 0), min(b + 1, H)
        xb = big[:, ra:rb]                                   # row bands are pixel-dense views, < 2 GiB: DMA kernel
        dyb = dy[:, ra:rb].clone()
        if ra < a:
            dyb[:, 0] = 0                                    # halo rows contribute x only
        if rb > b:
            dyb[:, -1] = 0
        part = torch.empty_like(full)
        ops.conv3x3_wgrad(dyb, xb, None, part)
        acc += part
    # (the row bands take the LDS-DMA kernel, whose per-split partial sums travel as block-scaled fp16 -- 2^-12 each; the 2 GiB
    # tensor takes the register-staged kernel with fp32 slabs)
    assert _rel(full, acc) < 1e-3, _rel(full, acc)


def test_batchnorm_relu_kernels_at_2gib(big):
    from unet_amd import ops
    from unet_amd._lib import LIB, UH_BF16
    dev = big.device
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(2)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.3).to(dev)
    mean = (torch.randn(C, generator=g) * 0.1).to(dev)
    rstd = (torch.rand(C, generator=g) + 0.7).to(dev)
    scale = (gamma * rstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    n = H * W
    y = big
    z = torch.empty_like(y)
    LIB.call("uh_bn_relu_apply", y.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), z.data_ptr(), C, n, C, UH_BF16, st)
    gd = torch.Generator(device=dev).manual_seed(8)
    dz = torch.randn(1, H, W, C, device=dev, dtype=torch.bfloat16, generator=gd)
    nblk = LIB.query("uh_bn_bwd_nblk", n, C)
    partials = torch.empty(nblk * 2 * C, dtype=torch.float32, device=dev)
    dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C, dtype=torch.float32, device=dev)
    dy = torch.empty_like(y)
    LIB.call("uh_bn_relu_bwd_reduce", dz.data_ptr(), C, y.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
             rstd.data_ptr(), partials.data_ptr(), n, C, UH_BF16, st)
    LIB.call("uh_bn_relu_bwd_apply", dz.data_ptr(), C, y.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
             rstd.data_ptr(), partials.data_ptr(), nblk, dgamma.data_ptr(), dbeta.data_ptr(), dy.data_ptr(), C, n, 0, C,
             UH_BF16, st)
    torch.cuda.synchronize()
    # stock PyTorch on row bands (fp32 / fp64 accumulators)
    sg = torch.zeros(C, dtype=torch.float64, device=dev)
    sgx = torch.zeros(C, dtype=torch.float64, device=dev)
    for r in range(0, H, 256):
        yb = y[0, r:r + 256].float()
        zb = torch.relu(yb * scale + shift)
        assert _rel(z[0, r:r + 256], zb) < 1e-2
        gb = dz[0, r:r + 256].float() * (zb > 0)
        xh = (yb - mean) * rstd
        sg += gb.double().sum((0, 1))
        sgx += (gb * xh).double().sum((0, 1))
    assert _rel(dbeta, sg) < 1e-4 and _rel(dgamma, sgx) < 1e-4
    for r in (0, 1792, H - 256):
        yb = y[0, r:r + 256].float()
        gb = dz[0, r:r + 256].float() * ((yb * scale + shift) > 0)
        xh = (yb - mean) * rstd
        ref = scale * (gb - (sg / n).float() - xh * (sgx / n).float())
        assert _rel(dy[0, r:r + 256], ref) < 1e-2


def test_maxpool_at_2gib(big):
    from unet_amd import ops
    x = big.detach().requires_grad_(True)
    y = ops.MaxPool2Fn.apply(x)
    gd = torch.Generator(device=big.device).manual_seed(9)
    dy = torch.randn(y.shape, device=big.device, dtype=torch.bfloat16, generator=gd)
    (dx,) = torch.autograd.grad(y, x, dy)
    torch.cuda.synchronize()
    assert y.shape == (1, H // 2, W // 2, C) and dx.shape == x.shape
    for r in (0, H - 512):                                   # stock PyTorch on the CPU, one band of rows at a time
        xb = big[0, r:r + 512].float().cpu().permute(2, 0, 1)[None].requires_grad_(True)
        yb = F.max_pool2d(xb, 2)
        assert torch.equal(y[0, r // 2:r // 2 + 256].float().cpu().permute(2, 0, 1)[None], yb.detach())
        (dxb,) = torch.autograd.grad(yb, xb, dy[0, r // 2:r // 2 + 256].float().cpu().permute(2, 0, 1)[None])
        assert torch.equal(dx[0, r:r + 512].float().cpu().permute(2, 0, 1)[None], dxb)


def test_upsample_to_2gib_samples_and_adjoint():
    from unet_amd import ops
    dev = _dev()
    h = w = H // 2
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(1, h, w, C, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    u = ops.UpsampleBilinearPadFn.apply(x, H, W)
    assert u.shape == (1, H, W, C) and u.numel() * u.element_size() == 1 << 31
    # closed form of nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (unet_parts.py:69) at sampled pixels
    gi = torch.Generator().manual_seed(12)
    oy = torch.cat([torch.randint(0, H, (4000,), generator=gi), torch.tensor([0, H - 1, H - 1, 0, H // 2])]).to(dev)
    ox = torch.cat([torch.randint(0, W, (4000,), generator=gi), torch.tensor([0, W - 1, 0, W - 1, W // 2])]).to(dev)
    sy, sx = (h - 1) / (H - 1), (w - 1) / (W - 1)
    fy, fx = oy.double() * sy, ox.double() * sx
    y0, x0 = fy.floor().long().clamp(max=h - 1), fx.floor().long().clamp(max=w - 1)
    y1, x1 = (y0 + 1).clamp(max=h - 1), (x0 + 1).clamp(max=w - 1)
    ly, lx = (fy - y0).unsqueeze(1), (fx - x0).unsqueeze(1)
    xd = x.detach()[0].double()
    ref = (1 - ly) * ((1 - lx) * xd[y0, x0] + lx * xd[y0, x1]) + ly * ((1 - lx) * xd[y1, x0] + lx * xd[y1, x1])
    got = u.detach()[0][oy, ox].double()
    assert float((got - ref).abs().max()) < 4e-2 * float(ref.abs().max())      # bf16 output rounding + fp32 lerp weights
    # backward is the adjoint:  <U x, dy> = <x, U^T dy>
    gd = torch.Generator(device=dev).manual_seed(13)
    dy = torch.randn(1, H, W, C, device=dev, dtype=torch.bfloat16, generator=gd)
    (dx,) = torch.autograd.grad(u, x, dy)
    lhs = sq = 0.0
    for r in range(0, H, 256):
        t = u.detach()[0, r:r + 256].double() * dy[0, r:r + 256].double()
        lhs += float(t.sum())
        sq += float((t * t).sum())
    t = x.detach().double() * dx.double()
    rhs = float(t.sum())
    sq += float((t * t).sum())
    # u and dx are rounded to bf16 (2^-9 relative, independent per element): the two sums differ by a random walk
    assert abs(lhs - rhs) < 6.0 * 2.0 ** -8 * sq ** 0.5, (lhs, rhs, sq ** 0.5)


def test_transposed_conv_backward_takes_a_strided_gradient_past_2gib():
    """Up's ConvTranspose2d (unet_parts.py:73) receives its gradient as the [.., C:] half of the concatenated tensor's gradient
    (unet_parts.py:95).  At 32 x 512 x 512 x 64 bf16 channels that half is 1 GiB packed but a 2 GiB walk at the concatenated
    tensor's pixel stride -- past the MFMA kernels' buffer window (bench.py --convt stopped there in its global-batch-32 leg).  The
    op packs such a gradient first: the result must be bit-identical to the packed call."""
    from unet_amd import ops
    dev = _dev()
    Bn, hh, Cin, Cout = 32, 256, 128, 64
    g = torch.Generator(device=dev).manual_seed(21)
    x = torch.randn(Bn, hh, hh, Cin, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    wt = (torch.randn(Cin, Cout, 2, 2, device=dev, generator=g) * 0.05).requires_grad_(True)
    b = torch.zeros(Cout, device=dev).requires_grad_(True)
    cat = torch.randn(Bn, 2 * hh, 2 * hh, 2 * Cout, device=dev, dtype=torch.bfloat16, generator=g)     # 2 GiB: the concatenated gradient
    dy_strided = cat[..., Cout:]
    assert ops.pixel_ld(dy_strided) == 2 * Cout and cat.numel() * cat.element_size() == 1 << 31
    y = ops.ConvTranspose2x2PadFn.apply(x, wt, b, 2 * hh, 2 * hh)
    got = torch.autograd.grad(y, [x, wt, b], dy_strided, retain_graph=True)
    want = torch.autograd.grad(y, [x, wt, b], dy_strided.contiguous())
    for a, c in zip(got, want):
        assert torch.equal(a, c)
    # and the values are those of the transposed conv: one batch element against torch on the CPU
    xs = x.detach()[:1, :32, :32].float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    ws = wt.detach().bfloat16().float().cpu().requires_grad_(True)
    ys = F.conv_transpose2d(xs, ws, None, stride=2)
    assert float((y.detach()[:1, :64, :64].float().cpu().permute(0, 3, 1, 2) - ys).abs().max()) < 2e-2 * float(ys.abs().max())
