"""The network's first layer with its conv output RECOMPUTED instead of stored (ops.StemConvBnReluFn over uh_stem_*): the
forward is bit-identical to the stored-output path (same FMA order, same roundings, same statistics rows); the backward
differs only in the order in which per-workgroup partial sums are formed, so BatchNorm / filter gradients agree to fp32
round-off."""
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
    assert torch.equal(a["z"], b["z"]), f"forward differs in {int((a['z'] != b['z']).sum())} elements"
    assert torch.equal(a["rm"], b["rm"]) and torch.equal(a["rv"], b["rv"]) and a["nbt"] == b["nbt"] == 1

    def rel(u, v):
        return float((u.double() - v.double()).abs().max() / v.double().abs().max().clamp_min(1e-30))
    assert rel(b["db"], a["db"]) < 2e-5 and rel(b["dg"], a["dg"]) < 2e-5, (rel(b["db"], a["db"]), rel(b["dg"], a["dg"]))
    # dy is rounded to bf16 before the contraction in both paths; a last-bit difference in the two per-channel sums can move
    # individual roundings, never more
    assert float((b["dw"].double() - a["dw"].double()).norm() / a["dw"].double().norm()) < 1e-3


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
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]        # the forward pass is the same arithmetic
    assert abs(a["gn"] - b["gn"]) <= 1e-5 * a["gn"]
    assert float((a["g"] - b["g"]).norm() / a["g"].norm()) < 1e-4
    assert float((a["p"] - b["p"]).abs().max()) <= 2.5e-4                          # one RMSprop step moves an element by <= lr / sqrt(1 - alpha) = 1e-3


def test_fp32_and_wider_images_keep_the_stored_path():
    from unet_amd import ops
    dev = _dev()
    assert not ops.stem_recompute_ok(torch.zeros(1, 8, 8, 1, device=dev), 1, 64)                          # fp32 parity path
    assert not ops.stem_recompute_ok(torch.zeros(1, 8, 8, 3, device=dev, dtype=torch.bfloat16), 3, 64)    # default: single-channel images only
    assert not ops.stem_recompute_ok(torch.zeros(1, 8, 8, 1, device=dev, dtype=torch.bfloat16), 1, 128)
