"""Inference path on the GPU: fused conv + eval-BatchNorm + ReLU epilogue against stock fp64
maths and against the two-pass kernels, class-index masks (argmax / threshold) against torch, predict_img and
evaluate() against the CPU oracle at full UNet width, checkpoint round trip through the reference's wire format."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    return torch.device("cuda:0")


SHAPES = [
    # B, H, W, C0, C1, Cout         (MFMA NBW=1/2, two sources, partial tiles, stem v3, generic + fallback pass)
    (2, 32, 32, 64, 0, 128),
    (8, 64, 64, 64, 0, 128),
    (2, 17, 23, 64, 64, 64),
    (1, 40, 24, 128, 128, 256),
    (3, 19, 33, 1, 0, 64),
    (2, 20, 20, 3, 0, 64),
    (2, 12, 14, 8, 8, 16),
    (2, 9, 11, 3, 0, 8),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C0,C1,Cout", SHAPES)
def test_conv_affine_relu_fused(dtype, B, H, W, C0, C1, Cout):
    from unet_amd import ops
    from unet_amd._lib import LIB
    dev = _dev()
    g = torch.Generator().manual_seed(7 * B + H + C0 + Cout)
    Cin = C0 + C1
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.relu(F.conv2d(x.double(), w.double(), padding=1) * scale.double()[None, :, None, None]
                 + shift.double()[None, :, None, None])
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    x0 = xg[..., :C0]
    x1 = xg[..., C0:] if C1 else None
    wf, _ = ops.pack_w3x3(w.to(dev), dtype, False)
    sc, sh = scale.to(dev), shift.to(dev)
    z = torch.empty(B, H, W, Cout, dtype=dtype, device=dev)
    LIB.call("uh_conv3x3_fwd_affine_relu", x0.data_ptr(), C0, ops.pixel_ld(x0), None if x1 is None else x1.data_ptr(), C1,
             0 if x1 is None else ops.pixel_ld(x1), wf.data_ptr(), z.data_ptr(), Cout, Cout, sc.data_ptr(), sh.data_ptr(),
             B, H, W, ops._dt(x0), torch.cuda.current_stream().cuda_stream)
    got = z.float().cpu().permute(0, 3, 1, 2).double()
    tol = 2e-5 if dtype == torch.float32 else 1e-2        # bf16: one rounding of the output (8 mantissa bits)
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    assert err < tol, f"fused conv+affine+relu: rel err {err:.3e}"
    # and against the two-pass kernels (conv -> stored tensor -> scale/shift/ReLU): identical in fp32 up to the fma
    y, _, _ = ops.conv3x3_fwd(x0, x1, wf, Cout, False)
    z2 = torch.empty_like(y)
    LIB.call("uh_bn_relu_apply", y.data_ptr(), Cout, sc.data_ptr(), sh.data_ptr(), z2.data_ptr(), Cout, B * H * W, Cout,
             ops._dt(y), torch.cuda.current_stream().cuda_stream)
    d = (z.float() - z2.float()).abs().max().item() / max(z2.float().abs().max().item(), 1e-30)
    assert d < (1e-6 if dtype == torch.float32 else 1.6e-2), f"fused vs two-pass: {d:.3e}"


def test_argmax_and_threshold_match_torch():
    from unet_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    for (B, C, H, W) in [(2, 3, 17, 23), (1, 4, 64, 64), (3, 1, 5, 7), (1, 2, 1, 1)]:
        logits = torch.randn(B, C, H, W, generator=g)
        logits[0, :, 0, 0] = 0.25                         # exact tie -> first index
        if C > 2:
            logits[0, 2, H - 1, W - 1] = float("nan")     # NaN counts as the maximum, like torch
        ld = logits.to(dev).contiguous(memory_format=torch.channels_last)
        got = ops.argmax_classes(ld)
        ref = logits.argmax(dim=1)
        assert got.dtype == torch.int64 and got.shape == ref.shape
        assert torch.equal(got.cpu(), ref)
        got2 = ops.argmax_classes(logits.to(dev))          # NCHW-contiguous input takes the copy branch
        assert torch.equal(got2.cpu(), ref)
    x = torch.randn(3, 33, 35, generator=g)
    x[0, 0, 0] = 0.0
    t = ops.threshold_mask(x.to(dev))
    assert torch.equal(t.cpu(), (torch.sigmoid(x) > 0.5).float())


def _oracle_eval_logits(model, images, bilinear, depth=4):
    from oracle import unet_ref as U
    st = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    return U.unet_forward(images, st, bilinear, depth=depth, training=False)


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)


@pytest.mark.parametrize("bilinear", [True, False])
def test_predict_img_matches_oracle(bilinear):
    from PIL import Image
    import unet_amd
    dev = _dev()
    torch.manual_seed(5)
    model = unet_amd.UNet_S(1, 3, bilinear=bilinear)
    _randomize_bn(model, 6)
    model = model.to(dev)
    rng = np.random.default_rng(0)
    arr = (rng.random((70, 90)) * 255).astype(np.uint8)          # odd size: exercises the pad rule of Up
    img = Image.fromarray(arr, mode="L")
    got = unet_amd.predict_img(model, img, dev)
    assert got.shape == (70, 90) and got.dtype == np.int64
    x = torch.from_numpy(arr.astype(np.float32) / 255.0)[None, None]
    logits = _oracle_eval_logits(model, x, bilinear)
    ref = logits.argmax(dim=1).squeeze(0).numpy()
    top2 = logits.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).squeeze(0).numpy()
    safe = margin > 3e-2 * float(logits.abs().max())               # predict_img runs under bf16 autocast (predict.py:22)
    assert safe.mean() > 0.2
    assert (got == ref)[safe].all(), "class index differs where the top-2 margin is safe"
    assert (got != ref).mean() < 0.05
    vis = unet_amd.mask_to_image(got)
    assert vis.size == (90, 70) and set(np.unique(np.asarray(vis))) <= {0, 128, 255}


def test_full_width_eval_forward_fp32_vs_oracle():
    import unet_amd
    dev = _dev()
    torch.manual_seed(8)
    model = unet_amd.UNet(1, 1, bilinear=True)
    _randomize_bn(model, 9)
    g = torch.Generator().manual_seed(10)
    images = torch.rand(2, 1, 96, 80, generator=g)
    ref = _oracle_eval_logits(model, images, True)
    model = model.to(dev).eval()
    with torch.no_grad():
        got = model(images.to(dev)).float().cpu()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-3, f"eval logits rel err {err:.3e}"
    margin = ref.abs() > 1e-3 * ref.abs().max()
    assert ((got > 0) == (ref > 0))[margin].all()
    # packed-filter cache: a parameter change (through torch, and through the fused optimizer) must be seen
    with torch.no_grad():
        model.inc.double_conv[0].weight.mul_(0.5)
        got2 = model(images.to(dev)).float().cpu()
    st_ref = _oracle_eval_logits(model, images, True)
    assert (got2 - st_ref).abs().max().item() / st_ref.abs().max().item() < 1e-3
    stepper = unet_amd.TrainStepper(model, lr=1e-3, amp=False)
    model.train()
    stepper.step(images.to(dev), torch.randint(0, 3, (2, 96, 80), generator=g).to(dev))
    model.eval()
    with torch.no_grad():
        got3 = model(images.to(dev)).float().cpu()
    st_ref3 = _oracle_eval_logits(model, images, True)
    assert (got3 - st_ref3).abs().max().item() / st_ref3.abs().max().item() < 1e-3
    assert (got3 - got2).abs().max().item() > 0


def test_checkpoint_round_trip_on_gpu(tmp_path):
    import unet_amd
    dev = _dev()
    torch.manual_seed(11)
    a = unet_amd.UNet_T(1, 3, bilinear=False).to(dev)
    _randomize_bn(a, 12)
    path = unet_amd.save_checkpoint(a, str(tmp_path / "ck" / "checkpoint_epoch5.pth"), mask_values=[0, 128, 255])
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert raw["mask_values"] == [0, 128, 255]
    b = unet_amd.UNet_T(1, 3, bilinear=False).to(dev)
    mv = unet_amd.load_checkpoint(b, path, device=dev)
    assert mv == [0, 128, 255]
    x = torch.rand(1, 1, 32, 32).to(dev)
    a.eval(); b.eval()
    with torch.no_grad():
        assert torch.equal(a(x), b(x))


def test_eval_batchnorm_coefficients_are_cached_and_invalidated():
    """The inference path folds BatchNorm(eval) to scale/shift once per BatchNorm and reuses it; anything that changes a
    parameter or a running statistic (a train-mode forward writes them through raw pointers; load_state_dict; in-place
    edits) must drop the cache."""
    import unet_amd
    from unet_amd._lib import LIB
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    x = torch.rand(2, 1, 64, 64, device=dev)
    calls = []
    real = LIB.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)

    LIB.call = spy
    try:
        m.eval()
        with torch.no_grad():
            y0 = m(x).clone()
            n0 = calls.count("uh_bn_eval_coeffs")
            y1 = m(x).clone()
            assert calls.count("uh_bn_eval_coeffs") == n0, "coefficients were recomputed"
            assert n0 == 18 and torch.equal(y0, y1)
        m.train()
        m(x)                                   # rewrites the running statistics
        m.eval()
        with torch.no_grad():
            y2 = m(x).clone()
        assert calls.count("uh_bn_eval_coeffs") == 2 * n0
        assert not torch.equal(y0, y2)
        with torch.no_grad():
            m.inc.double_conv[1].running_var.mul_(4.0)          # an in-place edit bumps the version
            y3 = m(x).clone()
        assert calls.count("uh_bn_eval_coeffs") == 2 * n0 + 1 and not torch.equal(y2, y3)
        # the reference semantics, recomputed from scratch by stock PyTorch modules
        ref = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
        ref.load_state_dict(m.state_dict())
        ref.eval()
        with torch.no_grad():
            y4 = ref(x)
        assert torch.allclose(y3, y4, rtol=1e-5, atol=1e-6)
    finally:
        LIB.call = real
