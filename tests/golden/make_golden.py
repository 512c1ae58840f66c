#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Run in the build container only (it needs /root/reference, which never travels):

    python tests/golden/make_golden.py

It imports ``unet.unet_parts``, ``unet.unet_model``, ``utils.dice_score`` and
``utils.boundary_loss`` from /root/reference (read-only; nothing is copied), feeds them
seeded inputs and stores inputs + expected outputs as .npz.  ``train.py`` itself cannot be
imported (cv2 + two missing modules), so the step fixtures drive the imported reference
model/loss modules through the statement sequence of train.py:113-159 using stock
``torch.optim.RMSprop`` / ``clip_grad_norm_`` exactly as train.py:80-81,153-159 do.

Fixture list: G1 DoubleConv, G2 Down, G3 Up bilinear (+odd-size pad),
G4 Up convT, G5 OutConv, G6 Dice, G7 boundary_loss, G8 UNet_T 3-step trajectories,
G9 full UNet scalars, G10 eval-mode logits/masks, G11 depth-5 net from reference parts,
G12 utils/data_loading.BasicDataset items on synthetic PNG files,
G13 full-width config-4 / config-5 nets on small images (scalars + first logits),
G14 full UNet(1,1,bilinear=True) in eval mode at 2x1x512x512: logits, `logit > 0` masks, margin histogram, Dice,
G15 the reference under torch.autocast('cpu', bfloat16) (train.py:116,233: AMP is the CLI default) NEXT TO its own fp32
result on the same inputs: DoubleConv(64,128) fwd/bwd, UNet_T 3-step trajectory, UNet(1,1,True) 2x1x64x64 step 0,
G16 UNet_S(1,3,bilinear=False) -- the reference CLI's default model and class count (train.py:253,235) -- 3-step trajectory.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
sys.path.insert(0, REF)
from unet.unet_parts import DoubleConv, Down, Up, OutConv  # noqa: E402
from unet.unet_model import UNet, UNet_T, UNet_S           # noqa: E402
from utils.dice_score import dice_coeff, multiclass_dice_coeff, dice_loss  # noqa: E402
from utils.boundary_loss import boundary_loss              # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy().copy()   # copy: .numpy() aliases live module buffers


def sd_np(mod, prefix="sd."):
    return {prefix + k: npy(v) for k, v in mod.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


def module_fwd_bwd(mod, inputs, tag):
    """train-mode forward + backward with a seeded cotangent; records everything."""
    mod.train()
    rec = {}
    rec.update(sd_np(mod, "sd0."))
    xs = [x.clone().requires_grad_(True) for x in inputs]
    y = mod(*xs)
    g = torch.Generator().manual_seed(123)
    cot = torch.randn(y.shape, generator=g)
    y.backward(cot)
    for i, x in enumerate(xs):
        rec[f"x{i}"] = npy(x)
        rec[f"dx{i}"] = npy(x.grad)
    rec["y"] = npy(y)
    rec["cot"] = npy(cot)
    for k, p in mod.named_parameters():
        rec["grad." + k] = npy(p.grad)
    rec.update(sd_np(mod, "sd1."))   # BN buffers after the train-mode forward
    mod.eval()
    with torch.no_grad():
        rec["y_eval"] = npy(mod(*[x.detach() for x in xs]))
    save(tag, **rec)


def randomize_bn(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) + 0.5)


def g1_to_g5():
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(11)
    m = DoubleConv(3, 8); randomize_bn(m, 1)
    module_fwd_bwd(m, [torch.randn(2, 3, 16, 16, generator=g)], "g1_doubleconv_3_8")
    m = DoubleConv(4, 8, 6); randomize_bn(m, 2)
    module_fwd_bwd(m, [torch.randn(2, 4, 16, 16, generator=g)], "g1_doubleconv_4_8_mid6")
    m = DoubleConv(32, 64); randomize_bn(m, 12)
    module_fwd_bwd(m, [torch.randn(2, 32, 20, 24, generator=g)], "g1_doubleconv_32_64")
    m = Down(8, 16); randomize_bn(m, 3)
    module_fwd_bwd(m, [torch.randn(2, 8, 16, 16, generator=g)], "g2_down_8_16")
    m = Down(8, 16); randomize_bn(m, 13)
    module_fwd_bwd(m, [torch.randn(2, 8, 15, 19, generator=g)], "g2_down_8_16_odd")
    m = Up(16, 8, bilinear=True); randomize_bn(m, 4)
    module_fwd_bwd(m, [torch.randn(2, 8, 8, 8, generator=g), torch.randn(2, 8, 16, 16, generator=g)],
                   "g3_up_bilinear_16_8")
    m = Up(16, 8, bilinear=True); randomize_bn(m, 5)
    module_fwd_bwd(m, [torch.randn(2, 8, 8, 9, generator=g), torch.randn(2, 8, 17, 19, generator=g)],
                   "g3_up_bilinear_16_8_oddpad")
    m = Up(16, 8, bilinear=False); randomize_bn(m, 6)
    module_fwd_bwd(m, [torch.randn(2, 16, 8, 8, generator=g), torch.randn(2, 8, 16, 16, generator=g)],
                   "g4_up_convt_16_8")
    m = Up(16, 8, bilinear=False); randomize_bn(m, 16)
    module_fwd_bwd(m, [torch.randn(2, 16, 8, 9, generator=g), torch.randn(2, 8, 17, 19, generator=g)],
                   "g4_up_convt_16_8_oddpad")
    for ncls in (1, 4):
        m = OutConv(8, ncls)
        module_fwd_bwd(m, [torch.randn(2, 8, 16, 16, generator=g)], f"g5_outconv_8_{ncls}")


def g6_dice():
    g = torch.Generator().manual_seed(21)
    rec = {}
    p3 = torch.rand(3, 12, 10, generator=g)
    t3 = (torch.rand(3, 12, 10, generator=g) > 0.6).float()
    rec["p3"], rec["t3"] = npy(p3), npy(t3)
    rec["dice3_rbf_true"] = npy(dice_coeff(p3, t3, reduce_batch_first=True))
    rec["dice3_rbf_false"] = npy(dice_coeff(p3, t3, reduce_batch_first=False))
    rec["dice2"] = npy(dice_coeff(p3[0], t3[0]))
    rec["loss3"] = npy(dice_loss(p3, t3, multiclass=False))
    pr = p3.clone().requires_grad_(True)
    dice_loss(pr, t3, multiclass=False).backward()
    rec["loss3_grad"] = npy(pr.grad)
    p4 = torch.softmax(torch.randn(2, 4, 9, 11, generator=g), dim=1)
    t4 = F.one_hot(torch.randint(0, 4, (2, 9, 11), generator=g), 4).permute(0, 3, 1, 2).float()
    rec["p4"], rec["t4"] = npy(p4), npy(t4)
    rec["mdice_rbf_true"] = npy(multiclass_dice_coeff(p4, t4, reduce_batch_first=True))
    rec["mdice_rbf_false"] = npy(multiclass_dice_coeff(p4, t4, reduce_batch_first=False))
    rec["mloss"] = npy(dice_loss(p4, t4, multiclass=True))
    pr = p4.clone().requires_grad_(True)
    dice_loss(pr, t4, multiclass=True).backward()
    rec["mloss_grad"] = npy(pr.grad)
    # sets_sum == 0 branch (dice_score.py:16): empty prediction and empty target
    z = torch.zeros(2, 6, 6)
    rec["dice_zero_rbf_true"] = npy(dice_coeff(z, z, reduce_batch_first=True))
    rec["dice_zero_rbf_false"] = npy(dice_coeff(z, z, reduce_batch_first=False))
    pz = torch.stack([torch.zeros(6, 6), torch.rand(6, 6, generator=g)])
    tz = torch.stack([torch.zeros(6, 6), (torch.rand(6, 6, generator=g) > 0.5).float()])
    rec["pz"], rec["tz"] = npy(pz), npy(tz)
    rec["dice_halfzero_rbf_false"] = npy(dice_coeff(pz, tz, reduce_batch_first=False))
    save("g6_dice", **rec)


def g7_boundary():
    g = torch.Generator().manual_seed(31)
    rec = {}

    def case(tag, pred, target, **kw):
        rec[tag + ".pred"] = npy(pred)
        rec[tag + ".target"] = npy(target)
        rec[tag + ".kw"] = np.array([kw.get("edge_width", 64), kw.get("edge_weight", 5.0)], dtype=np.float64)
        rec[tag + ".loss"] = npy(boundary_loss(pred, target, **kw))

    B, H, W = 2, 128, 128
    logits = torch.randn(B, H, W, generator=g) * 2
    t01 = (torch.rand(B, H, W, generator=g) > 0.5).float()
    case("train_style", logits, t01, edge_width=51, edge_weight=15)            # target in {0,1}
    t255 = torch.where(torch.rand(B, H, W, generator=g) > 0.5, 255.0, 128.0)
    t255[:, :30] = 0
    case("coded255", logits, t255, edge_width=51, edge_weight=15)
    big = torch.randn(B, H, W, generator=g) * 8                                 # |logit| > 10 -> sigmoid branch
    assert big.abs().max() > 10
    case("sigmoid_branch", big, t255, edge_width=20, edge_weight=5.0)
    case("interior_empty", logits, t255, edge_width=64, edge_weight=5.0)        # ew >= H/2
    case("edge_zero", logits, t255, edge_width=0, edge_weight=5.0)
    p4 = torch.randn(B, 4, 64, 48, generator=g) * 3
    t4 = torch.where(torch.rand(B, 64, 48, generator=g) > 0.4, 255.0, 0.0)
    case("fourd_c4", p4, t4, edge_width=10, edge_weight=7)
    p1 = torch.rand(B, 1, 40, 56, generator=g)                                  # probabilities, C == 1
    t1 = torch.where(torch.rand(B, 40, 56, generator=g) > 0.5, 255.0, 0.0)
    case("fourd_c1_prob", p1, t1, edge_width=7, edge_weight=3.0)
    pr = torch.rand(3, 33, 47, generator=g)                                     # odd sizes, B=3
    tr = torch.where(torch.rand(3, 33, 47, generator=g) > 0.5, 255.0, 0.0)
    case("odd_b3", pr, tr, edge_width=5, edge_weight=2.0)
    save("g7_boundary", **rec)


def ref_train_steps(model, images_list, masks_list, n_classes, lr=1e-5, record_grads=True, tag=None,
                    boundary_mc=0.0):
    """Statement sequence of train.py:113-159 on the imported reference model/loss modules, fp32."""
    opt = torch.optim.RMSprop(model.parameters(), lr=lr, weight_decay=1e-8, momentum=0.999, foreach=True)
    crit = nn.CrossEntropyLoss() if n_classes > 1 else nn.BCEWithLogitsLoss()
    model.train()
    rec = {}
    rec.update(sd_np(model, "sd0."))
    for s, (images, masks) in enumerate(zip(images_list, masks_list)):
        true_masks = masks.clone()
        masks_pred = model(images)
        if n_classes == 1:
            true_masks //= 2
            bce = crit(masks_pred.squeeze(1), true_masks.float())
            dl = dice_loss(torch.sigmoid(masks_pred.squeeze(1)), true_masks.float(), multiclass=False)
            bl = boundary_loss(masks_pred.squeeze(1), true_masks.float(), edge_width=51, edge_weight=15)
            loss = bce + dl + 0.25 * bl
            rec[f"s{s}.bce"] = npy(bce)
        else:
            ce = crit(masks_pred, true_masks)
            dl = dice_loss(F.softmax(masks_pred, dim=1).float(),
                           F.one_hot(true_masks, n_classes).permute(0, 3, 1, 2).float(), multiclass=True)
            loss = ce + dl
            rec[f"s{s}.ce"] = npy(ce)
            bl = torch.zeros(())
            if boundary_mc:
                bl = boundary_loss(masks_pred, true_masks.float(), edge_width=51, edge_weight=7)
                loss = loss + boundary_mc * bl
        assert not torch.isnan(loss).any()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        rec[f"s{s}.dice"] = npy(dl)
        rec[f"s{s}.boundary"] = npy(bl)
        rec[f"s{s}.loss"] = npy(loss)
        rec[f"s{s}.grad_norm"] = npy(gn)
        if record_grads:
            rec[f"s{s}.logits"] = npy(masks_pred)
            if s == 0:
                for k, p in model.named_parameters():
                    rec[f"s{s}.grad." + k] = npy(p.grad)   # after clipping
        opt.step()
        if record_grads and s == len(images_list) - 1:
            rec.update(sd_np(model, f"sd{s + 1}."))     # final state only (keeps fixtures small)
    return rec


def synth_batch(seed, b, c, h, w, nmask=3):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, c, h, w, generator=g), torch.randint(0, nmask, (b, h, w), generator=g)


def g8_unet_t():
    for bilinear in (True, False):
        torch.manual_seed(0)
        m = UNet_T(1, 1, bilinear=bilinear)
        batches = [synth_batch(100 + s, 2, 1, 64, 64) for s in range(3)]
        rec = ref_train_steps(m, [b[0] for b in batches], [b[1] for b in batches], 1, lr=1e-5)
        for s, (im, mk) in enumerate(batches):
            rec[f"s{s}.images"], rec[f"s{s}.masks"] = npy(im), npy(mk)
        save(f"g8_unet_t_{'bilinear' if bilinear else 'convt'}", **rec)
    # multi-class (CE + multiclass Dice) on UNet_T(3, 4)
    torch.manual_seed(0)
    m = UNet_T(3, 4, bilinear=True)
    batches = [synth_batch(200 + s, 2, 3, 64, 64, nmask=4) for s in range(2)]
    rec = ref_train_steps(m, [b[0] for b in batches], [b[1] for b in batches], 4, lr=1e-5)
    for s, (im, mk) in enumerate(batches):
        rec[f"s{s}.images"], rec[f"s{s}.masks"] = npy(im), npy(mk)
    save("g8_unet_t_multiclass", **rec)


def g9_full_unet():
    """cfg-1 scalars: UNet(1,1,bilinear=True), 2x1x512x512, 3 steps, lr 1e-5."""
    torch.manual_seed(0)
    m = UNet(1, 1, bilinear=True).to(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(1)
    images = torch.rand(2, 1, 512, 512, generator=g)
    masks = torch.randint(0, 3, (2, 512, 512), generator=g)
    rec = ref_train_steps(m, [images] * 3, [masks] * 3, 1, lr=1e-5, record_grads=False)
    rec = {k: v for k, v in rec.items() if not k.startswith("sd")}
    rec["note"] = np.array("weights torch.manual_seed(0); data Generator(1): rand(2,1,512,512), randint(0,3)")
    save("g9_unet_full_scalars", **rec)


def g10_eval():
    torch.manual_seed(3)
    for bilinear in (True, False):
        m = UNet_T(1, 1, bilinear=bilinear); randomize_bn(m, 9)
        m.eval()
        im, mk = synth_batch(300, 2, 1, 64, 64)
        with torch.no_grad():
            logits = m(im)
        rec = sd_np(m, "sd.")
        rec["images"], rec["masks"] = npy(im), npy(mk)
        rec["logits"] = npy(logits)
        rec["mask_pred"] = npy(logits.squeeze(1) > 0)
        rec["abs_margin_min"] = npy(logits.abs().min())
        t = (mk // 2).float()
        rec["dice"] = npy(dice_coeff((torch.sigmoid(logits.squeeze(1)) > 0.5).float(), t, reduce_batch_first=False))
        save(f"g10_eval_unet_t_{'bilinear' if bilinear else 'convt'}", **rec)


class Depth5(nn.Module):
    """cfg-4 topology composed from the reference's own unet_parts classes."""

    def __init__(self, n_channels, n_classes, w):
        super().__init__()
        self.inc = DoubleConv(n_channels, w[0])
        self.down1 = Down(w[0], w[1]); self.down2 = Down(w[1], w[2]); self.down3 = Down(w[2], w[3])
        self.down4 = Down(w[3], w[4]); self.down5 = Down(w[4], w[5] // 2)
        self.up1 = Up(w[5], w[4] // 2, True); self.up2 = Up(w[4], w[3] // 2, True)
        self.up3 = Up(w[3], w[2] // 2, True); self.up4 = Up(w[2], w[1] // 2, True)
        self.up5 = Up(w[1], w[0], True)
        self.outc = OutConv(w[0], n_classes)

    def forward(self, x):
        x1 = self.inc(x); x2 = self.down1(x1); x3 = self.down2(x2); x4 = self.down3(x3)
        x5 = self.down4(x4); x6 = self.down5(x5)
        x = self.up1(x6, x5); x = self.up2(x, x4); x = self.up3(x, x3); x = self.up4(x, x2)
        x = self.up5(x, x1)
        return self.outc(x)


def g11_depth5():
    torch.manual_seed(5)
    w = [4, 8, 16, 32, 64, 128]
    m = Depth5(3, 4, w)
    batches = [synth_batch(400 + s, 2, 3, 64, 96, nmask=4) for s in range(2)]
    rec = ref_train_steps(m, [b[0] for b in batches], [b[1] for b in batches], 4, lr=1e-5, boundary_mc=0.2)
    for s, (im, mk) in enumerate(batches):
        rec[f"s{s}.images"], rec[f"s{s}.masks"] = npy(im), npy(mk)
    rec["widths"] = np.array(w)
    save("g11_depth5_multiclass", **rec)


def g12_data_loading():
    """utils/data_loading.py: three synthetic grey images + {0,128,255} masks written as PNG, read back through the
    reference's BasicDataset (augment on, scale 1.0 and 0.5).  Stores the raw arrays and every item."""
    import tempfile
    from PIL import Image
    from utils.data_loading import BasicDataset
    rng = np.random.default_rng(12)
    rec = {}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "imgs")); os.makedirs(os.path.join(d, "masks"))
        names = ["a01", "b02", "c03"]
        for k, n in enumerate(names):
            h, w = 20 + 2 * k, 24 + 4 * k
            img = (rng.random((h, w)) * 255).astype(np.uint8)
            mask = rng.choice(np.array([0, 128, 255], np.uint8), size=(h, w), p=[0.2, 0.5, 0.3])
            Image.fromarray(img, mode="L").save(os.path.join(d, "imgs", n + ".png"))
            Image.fromarray(mask, mode="L").save(os.path.join(d, "masks", n + "_mask.png"))
            rec[f"raw.{n}.img"], rec[f"raw.{n}.mask"] = img, mask
        for scale in (1.0, 0.5):
            ds = BasicDataset(os.path.join(d, "imgs"), os.path.join(d, "masks"), scale, augment=True)
            order = list(ds.ids)
            rec[f"s{scale}.len"] = np.array(len(ds))
            rec[f"s{scale}.mask_values"] = np.array(ds.mask_values)
            for name in names:
                base = order.index(name) * 4
                for r in range(4):
                    it = ds[base + r]
                    rec[f"s{scale}.{name}.r{r}.image"] = npy(it["image"])
                    rec[f"s{scale}.{name}.r{r}.mask"] = npy(it["mask"])
        ds = BasicDataset(os.path.join(d, "imgs"), os.path.join(d, "masks"), 1.0, augment=False)
        rec["noaug.len"] = np.array(len(ds))
    save("g12_data_loading", **rec)


def g13_full_width():
    """Full-width nets on small images (weights from torch.manual_seed, so no state dict is stored): BASELINE config 5's
    UNet(1,1,bilinear=False) on 2x1x64x64 and config 4's depth-5 bilinear net (64..2048) on 1x3x64x64 with CE + multiclass
    Dice + 0.2 * boundary (4-D path); 2 steps each, logits of step 0 + every loss term and the gradient norm."""
    rec = {}
    torch.manual_seed(0)
    m = UNet(1, 1, bilinear=False)
    batches = [synth_batch(500 + s, 2, 1, 64, 64) for s in range(2)]
    r = ref_train_steps(m, [b[0] for b in batches], [b[1] for b in batches], 1, lr=1e-5, record_grads=False)
    rec.update({"cfg5." + k: v for k, v in r.items() if not k.startswith("sd")})
    torch.manual_seed(0)
    m = UNet(1, 1, bilinear=False)
    m.train()
    with torch.no_grad():
        rec["cfg5.s0.logits"] = npy(m(batches[0][0]))
    torch.manual_seed(0)
    w = [64, 128, 256, 512, 1024, 2048]
    m = Depth5(3, 4, w)
    batches = [synth_batch(600 + s, 1, 3, 64, 64, nmask=4) for s in range(2)]
    r = ref_train_steps(m, [b[0] for b in batches], [b[1] for b in batches], 4, lr=1e-5, record_grads=False, boundary_mc=0.2)
    rec.update({"cfg4." + k: v for k, v in r.items() if not k.startswith("sd")})
    torch.manual_seed(0)
    m = Depth5(3, 4, w)
    m.train()
    with torch.no_grad():
        rec["cfg4.s0.logits"] = npy(m(batches[0][0]))
    rec["note"] = np.array("weights torch.manual_seed(0) then ctor; data synth_batch(500+s,2,1,64,64) / synth_batch(600+s,1,3,64,64,nmask=4)")
    save("g13_full_width", **rec)


def g14_eval_full_unet():
    """evaluate.py:43-66 on the full-width UNet at the benchmarked image size: eval-mode forward (running statistics),
    `sigmoid(logit) > 0.5` masks, per-image Dice.  Weights = torch.manual_seed(0) + ctor (not stored), BatchNorm affine /
    running statistics = randomize_bn(m, 14) so that eval mode is not the identity normalisation."""
    torch.manual_seed(0)
    m = UNet(1, 1, bilinear=True); randomize_bn(m, 14)
    m.eval()
    im, mk = synth_batch(1400, 2, 1, 512, 512)
    with torch.no_grad():
        # a freshly initialised head is all-positive: centre the logits on their median (a bias edit, stored below) so that
        # the `logit > 0` mask is a non-trivial pattern with pixels arbitrarily close to the threshold
        m.outc.conv.bias -= m(im).median()
        logits = m(im)
    a = logits.abs().squeeze(1)
    scale = float(a.max())
    edges = np.array([0.0] + [10.0 ** e for e in range(-8, 1)]) * scale          # |logit| / max|logit| decades
    rec = {"logits": npy(logits), "mask_pred_bits": np.packbits(npy(logits.squeeze(1) > 0).reshape(-1)),
           "outc_bias": npy(m.outc.conv.bias), "abs_max": np.array(scale), "abs_margin_min": npy(a.min()), "margin_edges": edges,
           "margin_hist": np.histogram(npy(a).reshape(-1), bins=edges)[0]}
    t = (mk // 2).float()
    rec["dice"] = npy(dice_coeff((torch.sigmoid(logits.squeeze(1)) > 0.5).float(), t, reduce_batch_first=False))
    rec["note"] = np.array("UNet(1,1,True): torch.manual_seed(0) + ctor, randomize_bn(m, 14), eval(); data synth_batch(1400,2,1,512,512)")
    save("g14_eval_full_unet_512", **rec)


def amp_ctx(amp):
    return torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp)          # train.py:116 on a CUDA-less host


def g15_bf16_doubleconv():
    """One DoubleConv(64,128) forward + backward, fp32 and under CPU bf16 autocast, same weights / input / cotangent."""
    g = torch.Generator().manual_seed(150)
    torch.manual_seed(15)
    m = DoubleConv(64, 128); randomize_bn(m, 15)
    m.train()
    x = torch.randn(2, 64, 24, 40, generator=g)
    cot = torch.randn(2, 128, 24, 40, generator=g)
    rec = sd_np(m, "sd0.")
    rec["x0"], rec["cot"] = npy(x), npy(cot)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    for tag, amp in (("ref32", False), ("ref16", True)):
        m.load_state_dict(sd0)
        m.zero_grad(set_to_none=True)
        xr = x.clone().requires_grad_(True)
        with amp_ctx(amp):
            y = m(xr)
        rec[tag + ".y_dtype"] = np.array(str(y.dtype))
        y.float().backward(cot)
        rec[tag + ".y"] = npy(y.float())
        rec[tag + ".dx0"] = npy(xr.grad)
        for k, p in m.named_parameters():
            rec[tag + ".grad." + k] = npy(p.grad)
        for k, v in m.state_dict().items():
            if "running" in k:
                rec[tag + ".sd1." + k] = npy(v)
    save("g15_bf16_doubleconv_64_128", **rec)


def ref_train_steps_amp(model, batches, n_classes, amp, lr=1e-5):
    """ref_train_steps with train.py:116's autocast around forward + loss (GradScaler is disabled on a CUDA-less host:
    scale(loss) is loss, unscale_ / update are no-ops -- train.py:84,154-159)."""
    opt = torch.optim.RMSprop(model.parameters(), lr=lr, weight_decay=1e-8, momentum=0.999, foreach=True)
    crit = nn.CrossEntropyLoss() if n_classes > 1 else nn.BCEWithLogitsLoss()
    model.train()
    rec = {}
    for s, (images, masks) in enumerate(batches):
        true_masks = masks.clone()
        with amp_ctx(amp):
            masks_pred = model(images)
            if n_classes == 1:
                true_masks //= 2
                first = crit(masks_pred.squeeze(1), true_masks.float())
                dl = dice_loss(torch.sigmoid(masks_pred.squeeze(1)), true_masks.float(), multiclass=False)
                bl = boundary_loss(masks_pred.squeeze(1), true_masks.float(), edge_width=51, edge_weight=15)
                loss = first + dl + 0.25 * bl
            else:
                first = crit(masks_pred, true_masks)
                dl = dice_loss(F.softmax(masks_pred, dim=1).float(),
                               F.one_hot(true_masks, n_classes).permute(0, 3, 1, 2).float(), multiclass=True)
                bl = torch.zeros(())
                loss = first + dl
        assert not torch.isnan(loss).any()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        rec[f"s{s}.{'bce' if n_classes == 1 else 'ce'}"] = npy(first.float())
        rec[f"s{s}.dice"] = npy(dl.float())
        rec[f"s{s}.boundary"] = npy(bl.float())
        rec[f"s{s}.loss"] = npy(loss.float())
        rec[f"s{s}.grad_norm"] = npy(gn)
        rec[f"s{s}.logits"] = npy(masks_pred.float())
        if s == 0:
            for k, p in model.named_parameters():
                rec["s0.grad." + k] = npy(p.grad)          # after clipping
        opt.step()
    rec.update(sd_np(model, f"sd{len(batches)}."))
    return rec


def g15_bf16_unet_t():
    """G8's UNet_T(1,1,bilinear) trajectory (same weights, same batches) under bf16 autocast, with the fp32 one beside it."""
    batches = [synth_batch(100 + s, 2, 1, 64, 64) for s in range(3)]
    rec = {}
    for tag, amp in (("ref32", False), ("ref16", True)):
        torch.manual_seed(0)
        m = UNet_T(1, 1, bilinear=True)
        if not rec:
            rec.update(sd_np(m, "sd0."))
        r = ref_train_steps_amp(m, batches, 1, amp)
        rec.update({tag + "." + k: v for k, v in r.items()})
    for s, (im, mk) in enumerate(batches):
        rec[f"s{s}.images"], rec[f"s{s}.masks"] = npy(im), npy(mk)
    save("g15_bf16_unet_t_bilinear", **rec)


def g15_bf16_unet_full():
    """UNet(1,1,bilinear=True) (weights torch.manual_seed(0) + ctor, not stored) on 2x1x64x64, step 0, fp32 and bf16
    autocast: logits, loss terms, gradient norm; per parameter tensor the L2 norms of the fp32 gradient and of
    (bf16 gradient - fp32 gradient) -- the reference's OWN bf16 error, the yardstick of the HIP bf16 path -- and the
    gradients themselves for the tensors of at most 2^15 elements."""
    g = torch.Generator().manual_seed(1)
    images = torch.rand(2, 1, 64, 64, generator=g)
    masks = torch.randint(0, 3, (2, 64, 64), generator=g)
    rec = {"images": npy(images), "masks": npy(masks)}
    grads = {}
    for tag, amp in (("ref32", False), ("ref16", True)):
        torch.manual_seed(0)
        m = UNet(1, 1, bilinear=True)
        r = ref_train_steps_amp(m, [(images, masks)], 1, amp)
        grads[tag] = {k[8:]: v for k, v in r.items() if k.startswith("s0.grad.")}
        rec.update({tag + "." + k: v for k, v in r.items() if k.startswith("s0.") and not k.startswith("s0.grad.")})
    names = list(grads["ref32"])
    rec["grad_names"] = np.array(names)
    rec["grad_l2.ref32"] = np.array([np.linalg.norm(grads["ref32"][k].astype(np.float64)) for k in names])
    rec["grad_l2.ref16"] = np.array([np.linalg.norm(grads["ref16"][k].astype(np.float64)) for k in names])
    rec["grad_l2.diff"] = np.array([np.linalg.norm(grads["ref16"][k].astype(np.float64) - grads["ref32"][k]) for k in names])
    for k in names:
        if grads["ref32"][k].size <= 1 << 15:
            rec["ref32.grad." + k], rec["ref16.grad." + k] = grads["ref32"][k], grads["ref16"][k]
    rec["note"] = np.array("UNet(1,1,True): torch.manual_seed(0) + ctor; data Generator(1): rand(2,1,64,64), randint(0,3); "
                           "gradients are AFTER clip_grad_norm_(1.0), as train.py:157 leaves them")
    save("g15_bf16_unet_full_64", **rec)


def g16_unet_s():
    """train.py:253,235: the CLI's default model UNet_S(n_channels=1, n_classes=3, bilinear=False); CE + multiclass Dice
    (train.py:136-142), 3 steps, fp32.  Weights = torch.manual_seed(0) + ctor (not stored; per-tensor sums are)."""
    torch.manual_seed(0)
    m = UNet_S(1, 3, bilinear=False)
    init = {k: v.detach().clone() for k, v in m.state_dict().items()}
    batches = [synth_batch(700 + s, 2, 1, 64, 64, nmask=3) for s in range(3)]
    rec = ref_train_steps_amp(m, batches, 3, False)
    rec["sd0_names"] = np.array(list(init))
    rec["sd0_sums"] = np.array([float(v.double().sum()) for v in init.values()])
    rec["sd0_abs_sums"] = np.array([float(v.double().abs().sum()) for v in init.values()])
    for s, (im, mk) in enumerate(batches):
        rec[f"s{s}.images"], rec[f"s{s}.masks"] = npy(im), npy(mk)
    save("g16_unet_s_convt_3class", **rec)


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # python make_golden.py g14_eval_full_unet  -> that generator only
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_to_g5()
    g6_dice()
    g7_boundary()
    g8_unet_t()
    g10_eval()
    g11_depth5()
    g9_full_unet()
    g12_data_loading()
    g13_full_width()
    g14_eval_full_unet()
    g15_bf16_doubleconv()
    g15_bf16_unet_t()
    g15_bf16_unet_full()
    g16_unet_s()
