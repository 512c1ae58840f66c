"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI via the
drop-in modules, against (a) the golden fixtures produced by the reference's own modules and (b) the
CPU oracle (oracle/) on seeded inputs.  Tolerances: fp32 path 1e-3 relative to the tensor's max
(BASELINE.json north_star); bf16 path: no hand-set tolerance -- err(HIP bf16, reference fp32) against the reference's own
err(bf16 autocast, fp32) on the same inputs (tests/yardstick.py, tests/test_gpu_bf16_vs_reference.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; the product path has no CPU fallback")
    return torch.device("cuda:0")


def T(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def relerr(a, b):
    a = a.detach().double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


_REPORT = []


def check(a, b, tol, what="", atol=0.0, l2=False):
    """max |a-b| <= tol*max|b| + atol, or (l2=True) ||a-b||_2 <= tol*||b||_2 -- the right yardstick for the
    bf16 path, where single ReLU-mask flips near zero move individual elements by one whole term."""
    a64 = a.detach().double().cpu()
    b64 = torch.as_tensor(b).double().cpu()
    if l2:
        err = float((a64 - b64).norm())
        bound = tol * float(b64.norm().clamp_min(1e-12)) + atol
    else:
        err = float((a64 - b64).abs().max())
        bound = tol * float(b64.abs().max().clamp_min(1e-12)) + atol
    _REPORT.append(f"{what:50s} {'l2 ' if l2 else 'max'} err {err:.3e} bound {bound:.3e} ({err / max(bound, 1e-300):.2f} of bound)")
    assert err <= bound, f"{what}: {'l2' if l2 else 'max abs'} error {err:.3e} > {bound:.3e} (tol {tol:.1e}, atol {atol:.1e})"


@pytest.fixture(autouse=True, scope="module")
def _dump_report():
    yield
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.txt", "w") as f:
        f.write("\n".join(_REPORT) + "\n")


def load_sd(mod, rec, prefix):
    sd = {k[len(prefix):]: T(v) for k, v in rec.items() if k.startswith(prefix)}
    mod.load_state_dict(sd)


def run_block(mod, rec, dev, n_in, tol=1e-3):
    import unet_amd  # noqa: F401
    mod = mod.to(dev)
    load_sd(mod, rec, "sd0.")
    mod.train()
    xs = [T(rec[f"x{i}"], dev).requires_grad_(True) for i in range(n_in)]
    y = mod(*xs)
    check(y, rec["y"], tol, "fwd")
    y.backward(T(rec["cot"], dev))
    for i in range(n_in):
        check(xs[i].grad, rec[f"dx{i}"], tol, f"dx{i}")
    for k, p in mod.named_parameters():
        check(p.grad, rec["grad." + k], 2e-3, "grad " + k)
    sd1 = {k[4:]: v for k, v in rec.items() if k.startswith("sd1.")}
    for k, v in mod.state_dict().items():
        if "running" in k:
            check(v, sd1[k], 1e-4, k)
        if "num_batches" in k:
            assert int(v) == int(sd1[k])
    mod.eval()
    with torch.no_grad():
        check(mod(*[x.detach() for x in xs]), rec["y_eval"], tol, "eval fwd")


@pytest.mark.parametrize("name,args", [("g1_doubleconv_3_8", (3, 8)), ("g1_doubleconv_4_8_mid6", (4, 8, 6)),
                                       ("g1_doubleconv_32_64", (32, 64))])
def test_double_conv_golden(name, args):
    import unet_amd
    run_block(unet_amd.DoubleConv(*args), load_golden(name), _dev(), 1)


@pytest.mark.parametrize("name", ["g2_down_8_16", "g2_down_8_16_odd"])
def test_down_golden(name):
    import unet_amd
    run_block(unet_amd.Down(8, 16), load_golden(name), _dev(), 1)


@pytest.mark.parametrize("name,bilinear", [("g3_up_bilinear_16_8", True), ("g3_up_bilinear_16_8_oddpad", True),
                                           ("g4_up_convt_16_8", False), ("g4_up_convt_16_8_oddpad", False)])
def test_up_golden(name, bilinear):
    import unet_amd
    run_block(unet_amd.Up(16, 8, bilinear), load_golden(name), _dev(), 2)


@pytest.mark.parametrize("ncls", [1, 4])
def test_outconv_golden(ncls):
    import unet_amd
    run_block(unet_amd.OutConv(8, ncls), load_golden(f"g5_outconv_8_{ncls}"), _dev(), 1)


def test_dice_golden():
    import unet_amd
    dev = _dev()
    r = load_golden("g6_dice")
    p3, t3 = T(r["p3"], dev), T(r["t3"], dev)
    check(unet_amd.dice_coeff(p3, t3, True), r["dice3_rbf_true"], 1e-5)
    check(unet_amd.dice_coeff(p3, t3, False), r["dice3_rbf_false"], 1e-5)
    check(unet_amd.dice_coeff(p3[0], t3[0]), r["dice2"], 1e-5)
    pr = p3.clone().requires_grad_(True)
    loss = unet_amd.dice_loss(pr, t3)
    check(loss, r["loss3"], 1e-5)
    loss.backward()
    check(pr.grad, r["loss3_grad"], 1e-4)
    p4, t4 = T(r["p4"], dev), T(r["t4"], dev)
    check(unet_amd.multiclass_dice_coeff(p4, t4, True), r["mdice_rbf_true"], 1e-5)
    check(unet_amd.multiclass_dice_coeff(p4, t4, False), r["mdice_rbf_false"], 1e-5)
    pr = p4.clone().requires_grad_(True)
    unet_amd.dice_loss(pr, t4, multiclass=True).backward()
    check(pr.grad, r["mloss_grad"], 1e-4)
    z = torch.zeros(2, 6, 6, device=dev)
    check(unet_amd.dice_coeff(z, z, True), r["dice_zero_rbf_true"], 1e-6)
    check(unet_amd.dice_coeff(z, z, False), r["dice_zero_rbf_false"], 1e-6)
    check(unet_amd.dice_coeff(T(r["pz"], dev), T(r["tz"], dev), False), r["dice_halfzero_rbf_false"], 1e-5)
    with pytest.raises(AssertionError):
        unet_amd.dice_coeff(p3[0], t3[0], reduce_batch_first=True)
    with pytest.raises(AssertionError):
        unet_amd.dice_coeff(p3, t3[:, :5])


@pytest.mark.parametrize("case", ["train_style", "coded255", "sigmoid_branch", "interior_empty", "edge_zero",
                                  "fourd_c4", "fourd_c1_prob", "odd_b3"])
def test_boundary_golden(case):
    import unet_amd
    dev = _dev()
    r = load_golden("g7_boundary")
    ew, wt = r[case + ".kw"]
    got = unet_amd.boundary_loss(T(r[case + ".pred"], dev), T(r[case + ".target"], dev), edge_width=int(ew),
                                 edge_weight=float(wt))
    assert not got.requires_grad
    check(got, r[case + ".loss"], 2e-5, case)


def _run_traj(name, cls, args, n_classes, lr=1e-5, bmc=None, widths=None, seeded_init=None):
    import unet_amd
    dev = _dev()
    r = load_golden(name)
    if seeded_init is not None:
        # the fixture stores no initial state: weights = torch.manual_seed(seed) + ctor on the CPU (the drop-in classes create
        # their parameters in the reference's order with the reference's initialisers), checked against the stored sums
        torch.manual_seed(seeded_init)
        model = cls(*args)
        for k, v, want, scale in zip(model.state_dict(), model.state_dict().values(), r["sd0_sums"], r["sd0_abs_sums"]):
            assert k == str(r["sd0_names"][list(model.state_dict()).index(k)])
            assert abs(float(v.double().sum()) - want) <= 1e-9 * max(scale, 1.0), k
        model = model.to(dev)
    else:
        model = (unet_amd.UNetDepth(*args, widths=widths) if widths else cls(*args)).to(dev)
        load_sd(model, r, "sd0.")
    stepper = unet_amd.TrainStepper(model, lr=lr, amp=False)
    nsteps = sum(1 for k in r if k.endswith(".images"))
    for s in range(nsteps):
        im, mk = T(r[f"s{s}.images"], dev), T(r[f"s{s}.masks"], dev)
        model.train()
        terms = unet_amd.train_step(model, stepper.optimizer, im, mk, amp=False, boundary_weight=bmc)
        # later steps sit behind RMSprop's sign-like first updates (+-10*lr per weight, flipped by round-off on
        # near-zero gradients) and BatchNorm over as few as 32 samples at the bottleneck: chaotic at the 1e-2 level
        tl = 1e-3 if s == 0 else 2e-2
        check(terms["logits"], r[f"s{s}.logits"], tl, f"logits s{s}")
        check(terms["loss"], r[f"s{s}.loss"], 1e-4 if s == 0 else 3e-3, f"loss s{s}")
        check(terms["dice"], r[f"s{s}.dice"], 1e-4 if s == 0 else 3e-3, f"dice s{s}")
        if "bce" in terms:
            check(terms["bce"], r[f"s{s}.bce"], 1e-4 if s == 0 else 3e-3, f"bce s{s}")
        if "ce" in terms:
            check(terms["ce"], r[f"s{s}.ce"], 1e-4 if s == 0 else 3e-3, f"ce s{s}")
        check(terms["boundary"], r[f"s{s}.boundary"], 1e-4 if s == 0 else 2e-2, f"boundary s{s}")
        check(terms["grad_norm"], r[f"s{s}.grad_norm"], 1e-3 if s == 0 else 3e-2, f"grad_norm s{s}")
        if s == 0:
            named = dict(model.named_parameters())
            for k, p in named.items():
                g = stepper.optimizer.grad_of(p)       # clipped gradient, as the fixture stores it
                check(g, r[f"s0.grad.{k}"], 2e-2, "grad " + k, l2=True)   # conditioning: see test_full_unet_step_vs_oracle_fp32
    final = {k[len(f"sd{nsteps}."):]: v for k, v in r.items() if k.startswith(f"sd{nsteps}.")}
    for k, v in model.state_dict().items():
        if "num_batches" in k:
            assert int(v) == int(final[k])
        else:
            # parameters moved by 3 sign-like RMSprop steps of size ~10*lr: round-off on near-zero
            # gradients may flip individual updates (same allowance as tests/test_oracle_golden.py)
            # momentum 0.999 accumulates the +-10 sign-like steps: 10 + 20 + 30 = 60*lr of travel in 3 steps, so one
            # weight whose tiny gradient flips sign in all three steps ends up to 120*lr away
            check(v, final[k], 5e-3, "final " + k, atol=0.0 if "running" in k else 150 * lr)


def test_unet_t_bilinear_trajectory():
    import unet_amd
    _run_traj("g8_unet_t_bilinear", unet_amd.UNet_T, (1, 1, True), 1)


def test_unet_t_convt_trajectory():
    import unet_amd
    _run_traj("g8_unet_t_convt", unet_amd.UNet_T, (1, 1, False), 1)


def test_unet_t_multiclass_trajectory():
    import unet_amd
    _run_traj("g8_unet_t_multiclass", unet_amd.UNet_T, (3, 4, True), 4)


def test_unet_s_default_model_trajectory():
    """Fixture G16: UNet_S(1, 3, bilinear=False), the model and class count the reference's CLI builds by default
    (train.py:253,235): CE + multiclass Dice, transposed-conv Up blocks, 16 ... 256 channels (the narrow-tensor kernels)."""
    import unet_amd
    _run_traj("g16_unet_s_convt_3class", unet_amd.UNet_S, (1, 3, False), 3, seeded_init=0)


def test_depth5_multiclass_trajectory():
    import unet_amd
    _run_traj("g11_depth5_multiclass", None, (3, 4, True), 4, bmc=0.2, widths=(4, 8, 16, 32, 64, 128))


@pytest.mark.parametrize("name,bilinear", [("g10_eval_unet_t_bilinear", True), ("g10_eval_unet_t_convt", False)])
def test_eval_masks_bit_exact(name, bilinear):
    import unet_amd
    dev = _dev()
    r = load_golden(name)
    model = unet_amd.UNet_T(1, 1, bilinear).to(dev)
    load_sd(model, r, "sd.")
    model.eval()
    with torch.no_grad():
        logits = model(T(r["images"], dev))
    check(logits, r["logits"], 1e-3, "eval logits")
    pred = (logits.squeeze(1) > 0).cpu().numpy()
    margin = np.abs(r["logits"]).squeeze(1)
    safe = margin > 1e-3 * np.abs(r["logits"]).max()
    assert (pred == r["mask_pred"])[safe].all(), "argmax mask differs where |logit| margin is safe"
    assert (pred != r["mask_pred"]).mean() < 1e-3
    batches = [{"image": T(r["images"]), "mask": T(r["masks"])}]
    dice, _, _ = unet_amd.evaluate(model, batches, dev, amp=False)
    check(dice, r["dice"], 2e-3, "dice")


# ------------------------------------------------------------------ MFMA kernels vs the CPU oracle
def _oracle_block(kind, st, xs, cot, bilinear=True, dtype=torch.float64):
    from oracle import unet_ref as U
    st = {("x." + k): (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
    keys = U.param_keys(st)
    work = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in st.items()}
    xs = [x.to(dtype).requires_grad_(True) for x in xs]
    if kind == "double_conv":
        y = U.double_conv(xs[0], work, "x", True, {})
    elif kind == "up":
        y = U.up(xs[0], xs[1], work, "x", bilinear, True, {})
    else:
        y = U.down(xs[0], work, "x", True, {})
    grads = torch.autograd.grad(y, xs + [work[k] for k in keys], cot.to(dtype))
    return y.detach(), grads[:len(xs)], {k[2:]: g for k, g in zip(keys, grads[len(xs):])}


def _bf16_block_against_the_reference_graph(tag, kind, st, xs, cot, y, dxs, named_grads, bilinear=True):
    """bf16 leg of the block tests: the yardstick is the same block from stock torch.nn modules (oracle/nn_ref.py: the
    reference's graph, bit-identical to fixture set G15 in both legs) run on the CPU in fp32 and under
    torch.autocast('cpu', bfloat16) on these very inputs -- tests/yardstick.py."""
    from oracle import nn_ref as N
    from yardstick import Collector
    cin_total = st[[k for k in st if k.endswith("double_conv.0.weight")][0]].shape[1]
    cout = st[[k for k in st if k.endswith("double_conv.3.weight")][0]].shape[0]
    if kind == "double_conv":
        blk = N.double_conv_module(cin_total, cout)
    else:
        blk = N.up_module(cin_total, cout, bilinear)
    y32, dx32, g32 = N.block_fwd_bwd(blk, st, xs, cot, amp=False)
    y16, dx16, g16 = N.block_fwd_bwd(blk, st, xs, cot, amp=True)
    c = Collector(tag)
    c.tensor("y", y, y32, y16)
    for i, dx in enumerate(dxs):
        c.tensor(f"dx{i}", dx, dx32[i], dx16[i])
    for k, g in named_grads.items():
        c.tensor("grad " + k, g, g32[k], g16[k])
    c.done()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, None)])
@pytest.mark.parametrize("cin,cout,h,w", [(64, 128, 40, 56), (128, 64, 33, 17), (32, 256, 16, 16)])
def test_double_conv_mfma_vs_oracle(dtype, tol, cin, cout, h, w):
    import unet_amd
    dev = _dev()
    torch.manual_seed(cin + cout)
    mod = unet_amd.DoubleConv(cin, cout)
    if dtype == torch.float32:
        mod = mod.to(memory_format=torch.channels_last)
    x = torch.randn(2, cin, h, w)
    cot = torch.randn(2, cout, h, w)
    st = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    mod = mod.to(dev).train()
    xg = x.to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == torch.bfloat16)):
        y = mod(xg)
    y.float().backward(cot.to(dev))
    tag = f"dconv{cin}-{cout} {str(dtype)[6:]} "
    if dtype == torch.bfloat16:
        # (round 4 held these gradients to a hand-set 1e-1 relative L2 and used 0.6-0.99 of it; fixture G15 shows where that
        # comes from: the reference's OWN bf16 gradients of such a block sit 5-9.5e-2 from its fp32 ones)
        _bf16_block_against_the_reference_graph(tag, "double_conv", st, [x], cot, y.float(), [xg.grad],
                                                {k: p.grad for k, p in mod.named_parameters()})
        return
    yo, dxo, go = _oracle_block("double_conv", st, [x], cot)
    check(y.float(), yo, tol, tag + "y")
    check(xg.grad, dxo[0], tol * 2, tag + "dx")
    for k, p in mod.named_parameters():
        check(p.grad, go[k], tol * 3, tag + "grad " + k)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, None)])
@pytest.mark.parametrize("bilinear", [True, False])
def test_up_mfma_two_sources_vs_oracle(dtype, tol, bilinear):
    import unet_amd
    dev = _dev()
    torch.manual_seed(5)
    mod = unet_amd.Up(128, 64, bilinear)
    cx1 = 64 if bilinear else 128
    x1 = torch.randn(2, cx1, 12, 19)
    x2 = torch.randn(2, 64, 25, 39)          # odd sizes: pad rule + partial tiles
    cot = torch.randn(2, 64, 25, 39)
    st = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    mod = mod.to(dev).train()
    a, b = x1.to(dev).requires_grad_(True), x2.to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == torch.bfloat16)):
        y = mod(a, b)
    y.float().backward(cot.to(dev))
    tag = f"up{'bil' if bilinear else 'ct'} {str(dtype)[6:]} "
    if dtype == torch.bfloat16:
        _bf16_block_against_the_reference_graph(tag, "up", st, [x1, x2], cot, y.float(), [a.grad, b.grad],
                                                {k: p.grad for k, p in mod.named_parameters()}, bilinear)
        return
    yo, dxo, go = _oracle_block("up", st, [x1, x2], cot, bilinear)
    check(y.float(), yo, tol, tag + "y")
    check(a.grad, dxo[0], tol * 2, tag + "dx1")
    check(b.grad, dxo[1], tol * 2, tag + "dx2")
    for k, p in mod.named_parameters():
        check(p.grad, go[k], tol * 3, tag + "grad " + k)


def test_full_unet_step_vs_oracle_fp32():
    """UNet(1,1,bilinear=True) at 2x1x64x64: MFMA conv / wgrad / dgrad kernels on every layer."""
    import unet_amd
    from oracle import step_ref as S
    dev = _dev()
    torch.manual_seed(0)
    model = unet_amd.UNet(1, 1, bilinear=True)
    g = torch.Generator().manual_seed(1)
    images = torch.rand(2, 1, 64, 64, generator=g)
    masks = torch.randint(0, 3, (2, 64, 64), generator=g)
    st = {k: v.detach().clone() for k, v in model.state_dict().items()}
    st64 = {k: (v.double() if v.is_floating_point() else v) for k, v in st.items()}
    _, _, info = S.train_step(st64, None, images.double(), masks, n_classes=1, bilinear=True)
    model = model.to(dev)
    stepper = unet_amd.TrainStepper(model, amp=False)
    terms = stepper.step(images.to(dev), masks.to(dev))
    check(terms["logits"], info["logits"], 1e-3, "logits")
    check(terms["loss"], info["loss"], 1e-4, "loss")
    check(terms["grad_norm"], info["grad_norm"], 2e-3, "grad_norm")
    coef = float(S.clip_coef(info["grad_norm"], 1.0))
    failures = []
    for k, p in model.named_parameters():
        try:
            # Whole-network gradients of this randomly initialised 18-BatchNorm UNet are ill-conditioned
            # w.r.t. fp32 round-off: 1-ulp (1e-7) relative perturbations of the conv weights move stock
            # PyTorch's own CPU gradients by up to ~1e-2 in relative L2 (tests/test_conditioning.py), because
            # single ReLU-mask flips at |z| ~ 1e-6 re-route gradient through the BatchNorm projections.  Every
            # kernel is pinned to <= 2e-5 op by op (tests/test_gpu_ops.py); here 3e-2 L2 bounds the composition.
            check(stepper.optimizer.grad_of(p), info["grads"][k] * coef, 3e-2, "unet fp32 grad " + k, l2=True)
        except AssertionError as e:
            failures.append(str(e))
    assert not failures, "\n".join(failures)


def test_full_unet_step_bf16_against_the_reference_graph_under_autocast():
    """One bf16 step of UNet(1,1,bilinear=True) on 2x1x96x64 against the same step of the reference's graph (oracle/nn_ref.py,
    stock torch.nn modules, stock RMSprop) in fp32, with that graph under torch.autocast('cpu', bfloat16) as the yardstick
    (tests/yardstick.py): logits, every loss term, the gradient norm, every parameter's clipped gradient."""
    import unet_amd
    from oracle import nn_ref as N
    from yardstick import Collector
    dev = _dev()
    torch.manual_seed(0)
    model = unet_amd.UNet(1, 1, bilinear=True)
    g = torch.Generator().manual_seed(1)
    images = torch.rand(2, 1, 96, 64, generator=g)
    masks = torch.randint(0, 3, (2, 96, 64), generator=g)
    st = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = {}
    for amp in (False, True):
        m = N.NNUNet(1, 1, True)
        m.load_state_dict(st)
        ref[amp] = N.NNStepper(m, amp=amp).step(images, masks)
    model = model.to(dev)
    stepper = unet_amd.TrainStepper(model, amp=True)
    terms = stepper.step(images.to(dev), masks.to(dev))
    c = Collector("UNet 96x64")
    c.tensor("logits", terms["logits"].float(), ref[False]["logits"], ref[True]["logits"])
    for q in ("bce", "dice", "boundary", "loss", "grad_norm"):
        # (boundary_loss counts thresholded pixels, 12 288 of them here: a handful within bf16 round-off of the threshold move it
        # -- and the loss that carries a quarter of it -- by whole counts)
        c.scalar(q, float(terms[q].detach()), ref[False][q], ref[True][q], floor=1e-2 if q in ("boundary", "loss") else None)
    for k, p in model.named_parameters():
        c.tensor("grad " + k, stepper.optimizer.grad_of(p), ref[False]["grads"][k], ref[True]["grads"][k])
    c.done()


def test_missing_gpu_tensor_fails_loudly():
    import unet_amd
    m = unet_amd.DoubleConv(3, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 3, 8, 8))


def test_nan_loss_raises_and_leaves_parameters_untouched():
    """train.py:149-151: a NaN loss is fatal.  The flag is read after backward has been enqueued (train.py docstring of
    train_step) but before the optimizer step, so the parameters must be bit-identical afterwards."""
    import unet_amd
    dev = _dev()
    torch.manual_seed(0)
    model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    stepper = unet_amd.TrainStepper(model, lr=1e-3, amp=False)
    before = {k: v.clone() for k, v in model.state_dict().items() if v.is_floating_point() and "running" not in k}
    images = torch.rand(2, 1, 32, 32).to(dev)
    images[0, 0, 3, 3] = float("nan")
    masks = torch.randint(0, 3, (2, 32, 32)).to(dev)
    with pytest.raises(RuntimeError, match="NaN loss"):
        stepper.step(images, masks)
    torch.cuda.synchronize()
    for k, v in before.items():
        assert torch.equal(model.state_dict()[k], v), k
    # and the stepper still works afterwards
    images = torch.rand(2, 1, 32, 32).to(dev)
    out = stepper.step(images, masks)
    assert torch.isfinite(out["loss"]).all()


def test_dice_after_training_matches_oracle():
    """"Dice vs ref" (BASELINE metric), like for like: 200 (+ 25 + 25) train steps (lr 1e-4) on synthetic ellipse batches from the
    SAME five initialisations by the CPU oracle and by the HIP path (fp32, bf16), then evaluate.py's Dice on a held-out batch.
    The recipe itself is chaotic on this tiny problem (the oracle slides from 0.92 to 0.74 on one of the five initialisations,
    profiles/r04_dice_chaos_oracle.txt), so distributions are compared, not single runs: medians within 0.02, and no HIP run
    further below the oracle's worst run than 0.15 -- both on the per-run mean over the 200 / 225 / 250-step checkpoints, where a
    dip at one checkpoint weighs a third; the medians at step 200 itself are held too."""
    import bench
    r = bench.dice_vs_ref()
    _REPORT.append(f"dice_vs_ref {r}")
    assert r["ref_cpu_fp32"] > 0.9 and r["ref_cpu_fp32_avg"] > 0.9, r                      # the task is learned
    for name in ("hip_fp32", "hip_bf16"):
        assert abs(r[name + "_avg"] - r["ref_cpu_fp32_avg"]) < 0.02, (name, r)
        assert abs(r[name] - r["ref_cpu_fp32"]) < 0.02, (name, r)
        assert min(r[name + "_avg_runs"]) >= min(r["ref_cpu_fp32_avg_runs"]) - 0.15, (name, r)
    # like for like: the HIP bf16 runs beside the reference's OWN bf16 runs (the torch.nn graph under torch.autocast('cpu',
    # bfloat16), train.py:116) from the same five initialisations -- medians within 0.03, worst run no more than 0.15 below theirs
    assert abs(r["hip_bf16_avg"] - r["ref_cpu_bf16_avg"]) < 0.03, r
    assert min(r["hip_bf16_avg_runs"]) >= min(r["ref_cpu_bf16_avg_runs"]) - 0.15, r


@pytest.mark.parametrize("B,H,W,scale,w_b", [
    (2, 64, 64, 1.0, 0.25),        # edge band covers the image (2 * 51 >= 64): interior empty; raw logits used as probabilities
    (2, 160, 144, 1.0, 0.25),      # interior and edge regions, no sigmoid
    (3, 130, 171, 8.0, 0.25),      # |logit| > 10 somewhere: the sigmoid branch of boundary_loss.py:28
    (8, 512, 512, 1.5, 0.25),      # the benchmarked extent (1024 partial rows)
    (2, 96, 96, 1.0, 0.0),         # no boundary term
])
def test_fused_binary_loss_is_bit_identical_to_the_separate_kernels(B, H, W, scale, w_b):
    """uh_seg_loss_binary_fused (three launches) against uh_bce_dice_sums + uh_boundary_loss_mask + uh_seg_loss_binary_finish
    (six): every loss term, the sums kept for the backward pass, the gradient of the logits and the NaN flag."""
    from unet_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + H)
    logits = (torch.randn(B, H, W, generator=g) * scale).to(dev)
    mask = torch.randint(0, 3, (B, H, W), generator=g).to(dev)
    res = []
    default = ops.FUSE_LOSS
    try:
        for fuse in (False, True):
            ops.FUSE_LOSS = fuse
            lg = logits.clone().requires_grad_(True)
            total, bce, dice, bnd, nan_flag = ops.SegLossBinaryFn.apply(lg, mask, 2, w_b, 51, 15.0, None, 1)
            assert (nan_flag is not None) == fuse
            total.backward()
            res.append((total.detach().clone(), bce.clone(), dice.clone(), bnd.clone(), lg.grad.clone(), nan_flag))
    finally:
        ops.FUSE_LOSS = default
    torch.cuda.synchronize()
    for a, b, name in zip(res[0][:5], res[1][:5], ("total", "bce", "dice", "boundary", "dlogits")):
        assert torch.equal(a, b), (name, a, b)
    assert float(res[1][5]) == 0.0
    # a NaN logit raises the flag (and nothing else changes shape)
    bad = logits.clone()
    bad[0, 3, 5] = float("nan")
    out = ops.SegLossBinaryFn.apply(bad, mask, 2, w_b, 51, 15.0, None, 1)
    assert float(out[4]) == 1.0 and bool(torch.isnan(out[0]))


def test_cc_loss_option_adds_value_only():
    """BASELINE config 5: connected_component_loss (train.py:124-132, commented out upstream) added to the loss VALUE; the
    parameters after the step are identical with and without it (it carries no gradient)."""
    import unet_amd
    dev = _dev()
    im, mk = unet_amd.ellipse_batch(2, 64, seed=9)
    outs = []
    for cc in (False, True):
        torch.manual_seed(0)
        model = unet_amd.UNet_T(1, 1, bilinear=False).to(dev)
        st = unet_amd.TrainStepper(model, lr=1e-4, amp=False, cc_loss=cc)
        t = st.step(im.to(dev), mk.to(dev))
        torch.cuda.synchronize()
        outs.append((float(t["loss"].detach()), float(t.get("cc", torch.zeros(()))), {k: v.clone() for k, v in model.state_dict().items()}))
    assert outs[1][1] >= 0.0
    assert abs(outs[1][0] - (outs[0][0] + outs[1][1])) < 1e-5
    for k, v in outs[0][2].items():
        assert torch.equal(v, outs[1][2][k]), k


@pytest.mark.parametrize("ctor,bilinear,amp", [("UNet_T", True, False), ("UNet_T", False, False), ("UNet_S", False, True), ("UNet", True, True)])
def test_train_step_is_bit_deterministic(ctor, bilinear, amp):
    """No float atomics anywhere on the path: two runs of the same two steps give bit-identical logits, gradients and
    parameters (split-K slabs, BatchNorm partial rows and Dice partials are all reduced in a fixed order)."""
    import unet_amd
    dev = _dev()
    im, mk = unet_amd.ellipse_batch(2, 64, seed=9)
    res = []
    for _ in range(2):
        torch.manual_seed(0)
        model = getattr(unet_amd, ctor)(1, 1, bilinear=bilinear).to(dev)
        st = unet_amd.TrainStepper(model, lr=1e-4, amp=amp)
        for _ in range(2):
            t = st.step(im.to(dev), mk.to(dev))
        torch.cuda.synchronize()
        res.append((t["logits"].clone(), st.optimizer.flat_g.clone(), st.optimizer.flat_p.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("ctor,bilinear,amp", [("UNet_T", True, False), ("UNet_S", False, True), ("UNet", True, True)])
def test_graph_captured_step_matches_eager(ctor, bilinear, amp):
    """GraphedTrainStepper (the whole step replayed from a HIP graph) == TrainStepper, bit for bit, including the BatchNorm
    buffers; the capture's warm-up steps must not count as training; a NaN batch raises and leaves the weights alone."""
    import unet_amd
    dev = _dev()
    im, mk = unet_amd.ellipse_batch(2, 64, seed=4)
    im2, mk2 = unet_amd.ellipse_batch(2, 64, seed=5)
    res = []
    for cls in (unet_amd.TrainStepper, unet_amd.GraphedTrainStepper):
        torch.manual_seed(0)
        model = getattr(unet_amd, ctor)(1, 1, bilinear=bilinear).to(dev)
        st = cls(model, lr=1e-4, amp=amp)
        for a, b in ((im, mk), (im2, mk2), (im, mk)):
            t = st.step(a.to(dev), b.to(dev))
        torch.cuda.synchronize()
        res.append((float(t["loss"].detach()), {k: v.clone() for k, v in model.state_dict().items()}, st, model))
    assert res[0][0] == res[1][0]
    for k, v in res[0][1].items():
        assert torch.equal(v, res[1][1][k]), k
    st, model = res[1][2], res[1][3]
    before = {k: v.clone() for k, v in model.state_dict().items() if v.is_floating_point() and "running" not in k}
    bad = im.clone(); bad[0, 0, 1, 1] = float("nan")
    with pytest.raises(RuntimeError, match="NaN loss"):
        st.step(bad.to(dev), mk.to(dev))
    for k, v in before.items():
        assert torch.equal(model.state_dict()[k], v), k


@pytest.mark.parametrize("amp,graphed", [(True, False), (False, False), (True, True)])
def test_batched_slab_reduction_is_bit_identical_to_the_per_layer_launches(amp, graphed):
    """ops.SlabBatch: the closing reductions of backward-weights wait for ONE launch behind the backward pass (channels_last
    filters: the gradients go straight into the optimizer's flat buffer).  Same arithmetic, same order: parameters, clipped
    gradients and loss are bit-identical to the step that reduces every layer at once -- eager and replayed from a graph."""
    import unet_amd
    from unet_amd import ops
    dev = _dev()
    im, mk = unet_amd.ellipse_batch(2, 96, seed=12)
    res = []
    default = ops.DEFER_SLABS
    try:
        for defer in (False, True):
            ops.DEFER_SLABS = defer
            torch.manual_seed(0)
            model = unet_amd.UNet(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
            st = (unet_amd.GraphedTrainStepper if graphed else unet_amd.TrainStepper)(model, lr=1e-4, amp=amp)
            for _ in range(3):
                t = st.step(im.to(dev), mk.to(dev))
            torch.cuda.synchronize()
            if defer:
                assert len(st._slabs._tables) == 1 and next(iter(st._slabs._tables.values()))[1] == 17 and not st._slabs.rows     # 17 MFMA layers queued, flushed
            res.append((float(t["loss"].detach()), st.optimizer.flat_p.clone(), st.optimizer.flat_g.clone()))
            st.close()
    finally:
        ops.DEFER_SLABS = default
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_full_unet_step_fp32_bf16x3_vs_oracle():
    """fp32_mode='bf16x3' (3x3 forward / backward-data products on the bf16 matrix pipe, fp32 everywhere else) against the
    CPU oracle: logits and loss well inside the 1e-3 parity bar; gradients like the exact fp32 path (ill-conditioned, L2)."""
    import unet_amd
    from oracle import step_ref as S
    dev = _dev()
    torch.manual_seed(3)
    model = unet_amd.UNet(1, 1, bilinear=False)
    g = torch.Generator().manual_seed(4)
    images = torch.rand(2, 1, 64, 64, generator=g)
    masks = torch.randint(0, 3, (2, 64, 64), generator=g)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    _, _, ref = S.train_step(state, None, images, masks, n_classes=1, bilinear=False)
    model = model.to(dev)
    st = unet_amd.TrainStepper(model, amp=False, fp32_mode="bf16x3")
    t = st.step(images.to(dev), masks.to(dev))
    torch.cuda.synchronize()
    check(t["logits"], ref["logits"], 2e-4, "bf16x3 logits")
    check(t["loss"], ref["loss"], 1e-4, "bf16x3 loss")
    check(t["grad_norm"], ref["grad_norm"], 3e-2, "bf16x3 grad norm")
    from unet_amd import ops
    assert ops.FP32_MODE == "exact", "a stepper's fp32 mode is a per-step setting: it must not leak into later calls of the process"
