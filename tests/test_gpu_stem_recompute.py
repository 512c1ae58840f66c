"""The network's first layer with its conv output RECOMPUTED instead of stored (ops.StemConvBnReluFn over uh_stem_*, the
recomputation on the matrix pipe: csrc/stem_mfma.hip).  Same roundings as the stored-output path (y rounded to bf16 before
BatchNorm, dy before the contraction); the MFMA adds the nine products of a conv output in its own order, so y may differ in
the last fp32 bit = one bf16 ulp on about one element in 10^4, and the sums are formed in another order: everything agrees
with the stored path to bf16 / fp32 round-off, nothing more is claimed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    return torch.device("cuda:0")


def _layer(cin, seed):
    import torch.nn as nn
    torch.manual_seed(seed)
    conv = nn.Conv2d(cin, 64, 3, padding=1, bias=False)
    bn = nn.BatchNorm2d(64)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64) + 0.5)
        bn.bias.copy_(torch.randn(64) * 0.2)
    return conv, bn


@pytest.mark.parametrize("B,H,W,cin", [(2, 64, 64, 1), (3, 50, 37, 1), (2, 400, 300, 1), (1, 16, 16, 1), (2, 70, 33, 1), (1, 9, 200, 1)])
def test_recomputed_stem_matches_the_stored_path(B, H, W, cin):
    from unet_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 100 + H + cin)
    x = torch.rand(B, H, W, cin, generator=g).to(dev, torch.bfloat16)
    dz = torch.randn(B, H, W, 64, generator=g).to(dev, torch.bfloat16)
    res = {}
    old_max = ops.STEM_RECOMPUTE_MAX_CIN
    try:
        for mode in ("stored", "recomputed"):
            conv, bn = _layer(cin, 7)
            conv, bn = conv.to(dev), bn.to(dev)
            w = conv.weight
            if mode == "stored":
                z = ops.ConvBnReluFn.apply(x, None, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                           True, 0.1, bn.eps)
            else:
                assert ops.stem_recompute_ok(x, cin, 64)
                z = ops.StemConvBnReluFn.apply(x, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                               0.1, bn.eps)
            dw, dg, db = torch.autograd.grad(z, [w, bn.weight, bn.bias], dz)
            torch.cuda.synchronize()
            res[mode] = dict(z=z.detach().clone(), dw=dw.clone(), dg=dg.clone(), db=db.clone(), rm=bn.running_mean.clone(),
                             rv=bn.running_var.clone(), nbt=int(bn.num_batches_tracked))
    finally:
        ops.STEM_RECOMPUTE_MAX_CIN = old_max
    a, b = res["stored"], res["recomputed"]
    za, zb = a["z"].float(), b["z"].float()
    differ = float((za != zb).float().mean())
    assert differ < 2e-3, f"{differ:.2e} of the activations differ"
    assert float((za - zb).abs().max()) <= 2.0 ** -7 * float(za.abs().max())                 # and those by one bf16 ulp
    assert a["nbt"] == b["nbt"] == 1

    def rel(u, v):
        return float((u.double() - v.double()).abs().max() / v.double().abs().max().clamp_min(1e-30))
    assert rel(b["rm"], a["rm"]) < 1e-5 and rel(b["rv"], a["rv"]) < 1e-5
    assert rel(b["db"], a["db"]) < 2e-3 and rel(b["dg"], a["dg"]) < 2e-3, (rel(b["db"], a["db"]), rel(b["dg"], a["dg"]))
    assert float((b["dw"].double() - a["dw"].double()).norm() / a["dw"].double().norm()) < 2e-3


def test_recomputed_stem_against_fp64():
    """The four kernels against stock fp64 maths on the bf16-rounded operands: activation, BatchNorm parameter gradients,
    filter gradient (bf16 tolerances of the op tests)."""
    import torch.nn.functional as F
    from unet_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 70, 45
    x = torch.rand(B, H, W, 1, generator=g).bfloat16()
    dz = torch.randn(B, H, W, 64, generator=g).bfloat16()
    conv, bn = _layer(1, 9)
    with torch.no_grad():
        conv.weight.copy_(conv.weight.bfloat16().float())
    w64 = conv.weight.detach().double().requires_grad_(True)
    g64, b64 = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w64, padding=1)
    yq = y + (y.detach().float().bfloat16().double() - y.detach())           # straight-through rounding of the stored y
    mu, var = yq.mean((0, 2, 3), keepdim=True), yq.var((0, 2, 3), unbiased=False, keepdim=True)
    zr = torch.relu((yq - mu) / torch.sqrt(var + bn.eps) * g64[None, :, None, None] + b64[None, :, None, None])
    dwr, dgr, dbr = torch.autograd.grad(zr, [w64, g64, b64], dz.double().permute(0, 3, 1, 2))
    conv, bn = conv.to(dev), bn.to(dev)
    z = ops.StemConvBnReluFn.apply(x.to(dev), conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                   0.1, bn.eps)
    dw, dg, db = torch.autograd.grad(z, [conv.weight, bn.weight, bn.bias], dz.to(dev))
    torch.cuda.synchronize()

    def rel(u, v):
        return float((u.double().cpu() - v).abs().max() / v.abs().max())
    assert rel(z.permute(0, 3, 1, 2), zr.detach()) < 1e-2
    assert rel(dg, dgr) < 1e-2 and rel(db, dbr) < 1e-2
    assert float((dw.double().cpu() - dwr).norm() / dwr.norm()) < 2e-2


@pytest.mark.parametrize("bilinear", [True, False])
def test_train_step_with_and_without_the_recomputed_stem(bilinear):
    import unet_amd
    from unet_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    im = torch.rand(2, 1, 96, 96, generator=g).to(dev)
    mk = torch.randint(0, 3, (2, 96, 96), generator=g).to(dev)
    out = {}
    default = ops.STEM_RECOMPUTE
    try:
        for on in (False, True):
            ops.STEM_RECOMPUTE = on
            torch.manual_seed(0)
            model = unet_amd.UNet(1, 1, bilinear=bilinear).to(memory_format=torch.channels_last).to(dev)
            st = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
            t = st.step(im, mk)
            torch.cuda.synchronize()
            out[on] = dict(logits=t["logits"].clone(), loss=float(t["loss"]), gn=float(t["grad_norm"]), g=st.optimizer.flat_g.clone(),
                           p=st.optimizer.flat_p.clone())
            st.optimizer.close()
    finally:
        ops.STEM_RECOMPUTE = default
    a, b = out[False], out[True]
    # a bf16 ulp on a few first-layer activations travels through 17 more bf16 layers: bf16-level agreement, not equality
    assert float((a["logits"].float() - b["logits"].float()).norm() / a["logits"].float().norm()) < 4e-2      # measured 2.1e-2 (bf16 vs exact fp32: 6.5e-2)
    assert abs(a["loss"] - b["loss"]) <= 2e-3 * abs(a["loss"])
    assert abs(a["gn"] - b["gn"]) <= 5e-2 * a["gn"]
    # (per-element gradients of the whole bf16 network are not compared: a 2 x 96^2 batch leaves 72 samples per channel at the
    # bottleneck BatchNorm, and a one-ulp change of a first-layer activation re-routes ReLU masks all the way down -- the same
    # conditioning tests/test_conditioning.py shows for fp32; the kernels themselves are pinned by the two tests above)
    assert torch.isfinite(b["g"]).all() and torch.isfinite(b["p"]).all()


def test_fp32_and_wider_images_keep_the_stored_path():
    from unet_amd import ops
    dev = _dev()
    assert not ops.stem_recompute_ok(torch.zeros(1, 8, 8, 1, device=dev), 1, 64)                          # fp32 parity path
    assert not ops.stem_recompute_ok(torch.zeros(1, 8, 8, 3, device=dev, dtype=torch.bfloat16), 3, 64)    # default: single-channel images only
    assert not ops.stem_recompute_ok(torch.zeros(1, 8, 8, 1, device=dev, dtype=torch.bfloat16), 1, 128)
