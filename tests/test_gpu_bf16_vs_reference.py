"""The benchmarked arithmetic (bf16 activations, fp32 accumulation) against the REFERENCE'S OWN bf16: fixture set G15 holds what
the reference's modules produce under torch.autocast('cpu', bfloat16) -- /root/reference/train.py:116, and AMP is the CLI's
default (train.py:233) -- next to their fp32 result on the same inputs.  Criteria: tests/yardstick.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from yardstick import Collector

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; the product path has no CPU fallback")
    return torch.device("cuda:0")


def T(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def test_double_conv_block_against_the_reference_under_autocast():
    """DoubleConv(64, 128) forward + backward (unet_parts.py:7-24), MFMA kernels on both layers."""
    import unet_amd
    dev = _dev()
    r = load_golden("g15_bf16_doubleconv_64_128")
    assert str(r["ref16.y_dtype"]) == "torch.bfloat16"          # the reference's block really ran in bf16
    mod = unet_amd.DoubleConv(64, 128)
    mod.load_state_dict({k[4:]: T(v) for k, v in r.items() if k.startswith("sd0.")})
    mod = mod.to(dev).train()
    x = T(r["x0"], dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mod(x)
    assert y.dtype == torch.bfloat16
    y.float().backward(T(r["cot"], dev))
    c = Collector("DoubleConv(64,128)")
    c.tensor("y", y.float(), r["ref32.y"], r["ref16.y"])
    c.tensor("dx", x.grad, r["ref32.dx0"], r["ref16.dx0"])
    for k, p in mod.named_parameters():
        c.tensor("grad " + k, p.grad, r["ref32.grad." + k], r["ref16.grad." + k])
    for k, v in mod.state_dict().items():
        if "running" in k:
            c.tensor(k, v, r["ref32.sd1." + k], r["ref16.sd1." + k], floor=1e-3)
    c.done()


def test_unet_t_trajectory_against_the_reference_under_autocast():
    """G8's three steps of UNet_T(1,1,bilinear) (train.py:113-159) with amp=True: every loss term, the gradient norm, the
    logits of every step, every parameter's clipped gradient of step 0, and where the parameters are after step 3."""
    import unet_amd
    dev = _dev()
    r = load_golden("g15_bf16_unet_t_bilinear")
    model = unet_amd.UNet_T(1, 1, bilinear=True)
    model.load_state_dict({k[4:]: T(v) for k, v in r.items() if k.startswith("sd0.")})
    model = model.to(dev)
    lr = 1e-5
    stepper = unet_amd.TrainStepper(model, lr=lr, amp=True)
    c = Collector("UNet_T")
    for s in range(3):
        terms = stepper.step(T(r[f"s{s}.images"], dev), T(r[f"s{s}.masks"], dev))
        for q in ("bce", "dice", "boundary", "loss", "grad_norm"):
            c.scalar(f"s{s} {q}", float(terms[q].detach()), r[f"ref32.s{s}.{q}"], r[f"ref16.s{s}.{q}"], loose=s > 0)
        c.tensor(f"s{s} logits", terms["logits"].float(), r[f"ref32.s{s}.logits"], r[f"ref16.s{s}.logits"], loose=s > 0)
        if s == 0:
            for k, p in model.named_parameters():
                c.tensor("s0 grad " + k, stepper.optimizer.grad_of(p), r["ref32.s0.grad." + k], r["ref16.s0.grad." + k])
    # parameters: distance travelled from the initial state (3 sign-like RMSprop steps), so that the unchanged bulk of a
    # weight does not hide the update
    for k, p in model.named_parameters():
        p0 = r["sd0." + k].astype(np.float64)
        c.tensor("travel " + k, p.detach().double().cpu().numpy() - p0, r["ref32.sd3." + k] - p0, r["ref16.sd3." + k] - p0, loose=True)
    for k, v in model.state_dict().items():
        if "running" in k:
            c.tensor(k, v, r["ref32.sd3." + k], r["ref16.sd3." + k], floor=1e-3, loose=True)
    c.done()


def test_full_unet_step_against_the_reference_under_autocast():
    """UNet(1,1,bilinear=True) at 2x1x64x64, step 0: the full-width kernels.  The fixture stores the reference's per-tensor
    bf16 error (and the small gradients themselves); the fp32 gradients of the 17.3 M parameters are rebuilt here by the CPU
    oracle (pinned to the reference's fp32 at 1e-3 by G8 / G9 / G13) and checked against the stored norms first."""
    import unet_amd
    from oracle import step_ref as S
    dev = _dev()
    r = load_golden("g15_bf16_unet_full_64")
    torch.manual_seed(0)
    model = unet_amd.UNet(1, 1, bilinear=True)
    images, masks = T(r["images"]), T(r["masks"])
    st = {k: v.detach().clone() for k, v in model.state_dict().items()}
    _, _, info = S.train_step(st, None, images, masks, n_classes=1, bilinear=True)
    coef = float(S.clip_coef(info["grad_norm"], 1.0))
    names = [str(k) for k in r["grad_names"]]
    own = dict(zip(names, r["grad_l2.diff"] / r["grad_l2.ref32"]))
    for k, n32 in zip(names, r["grad_l2.ref32"]):
        assert abs(float(info["grads"][k].double().norm()) * coef - n32) <= 2e-2 * n32, k      # the oracle IS the fp32 reference
    model = model.to(dev)
    stepper = unet_amd.TrainStepper(model, amp=True)
    terms = stepper.step(images.to(dev), masks.to(dev))
    c = Collector("UNet 64x64")
    for q in ("bce", "dice", "boundary", "loss", "grad_norm"):
        c.scalar(q, float(terms[q].detach()), r[f"ref32.s0.{q}"], r[f"ref16.s0.{q}"])
    c.tensor("logits", terms["logits"].float(), r["ref32.s0.logits"], r["ref16.s0.logits"])
    from yardstick import FACTOR, FACTOR_LOOSE, l2
    for k, p in model.named_parameters():
        g = stepper.optimizer.grad_of(p)
        ref32 = r["ref32.grad." + k] if "ref32.grad." + k in r else (info["grads"][k] * coef).numpy()
        c.add("grad " + k, l2(g, ref32), float(own[k]), factor=FACTOR if p.numel() >= 1024 else FACTOR_LOOSE)
    c.done()
