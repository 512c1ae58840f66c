"""Oracle (test infrastructure): the reference's graph built from stock ``torch.nn`` modules.

``unet_ref`` restates the blocks as pure functions with a written-out BatchNorm; this file is the
other restatement SURVEY.md 8(d) asks the CPU baseline to be: "the same graph via torch.nn" --
``nn.Conv2d / nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d / nn.Upsample / nn.ConvTranspose2d`` wired as
/root/reference/unet/unet_parts.py:7-106 and /root/reference/unet/unet_model.py:8-38 wire them, with
the reference's ``state_dict`` keys, driven through the statement sequence of
/root/reference/train.py:113-159 with stock ``torch.optim.RMSprop`` (train.py:80-81),
``nn.BCEWithLogitsLoss`` / ``nn.CrossEntropyLoss`` (train.py:85), ``torch.autocast`` (train.py:116)
and ``clip_grad_norm_`` (train.py:157).  Under ``amp=True`` every op therefore takes the dtype the
reference's own modules take under CPU bf16 autocast (fixture set G15 pins that leg,
tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import losses_ref as L
from .unet_ref import unet_spec


def _conv_bn_relu_twice(cin: int, cout: int, mid: int) -> nn.Module:
    """unet_parts.py:14-21; wrapped so that the keys read ``double_conv.{0,1,3,4}.*``."""
    holder = nn.Module()
    holder.double_conv = nn.Sequential(
        nn.Conv2d(cin, mid, 3, padding=1, bias=False), nn.BatchNorm2d(mid), nn.ReLU(inplace=True),
        nn.Conv2d(mid, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
    return holder


def double_conv_module(cin: int, cout: int, mid: int = None) -> nn.Module:
    """DoubleConv(in, out, mid) (unet_parts.py:7-24) with the reference's keys (``double_conv.N.*``)."""
    blk = _conv_bn_relu_twice(cin, cout, mid or cout)
    blk.forward = lambda x: blk.double_conv(x)
    return blk


def down_module(cin: int, cout: int) -> nn.Module:
    """Down(in, out) (unet_parts.py:26-37), keys ``maxpool_conv.1.double_conv.N.*``."""
    blk = nn.Module()
    blk.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _conv_bn_relu_twice(cin, cout, cout))
    blk.forward = lambda x: blk.maxpool_conv[1].double_conv(blk.maxpool_conv[0](x))
    return blk


def up_module(cin: int, cout: int, bilinear: bool) -> nn.Module:
    """Up(in, out, bilinear) (unet_parts.py:62-98), keys ``up.*`` (transposed conv only) and ``conv.double_conv.N.*``."""
    blk = nn.Module()
    blk.up = (nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) if bilinear
              else nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2))
    blk.conv = _conv_bn_relu_twice(cin, cout, cin // 2 if bilinear else cout)

    def forward(x1, x2):
        y = blk.up(x1)
        dy, dx = x2.shape[2] - y.shape[2], x2.shape[3] - y.shape[3]
        y = F.pad(y, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return blk.conv.double_conv(torch.cat([x2, y], dim=1))
    blk.forward = forward
    return blk


def block_fwd_bwd(blk: nn.Module, state: Dict[str, torch.Tensor], xs, cot: torch.Tensor, amp: bool):
    """Train-mode forward + backward of one block on CPU, fp32 or under torch.autocast('cpu', bfloat16) (train.py:116):
    -> (y as fp32, [dx...], {name: parameter gradient})."""
    blk.load_state_dict(state)
    blk.train()
    blk.zero_grad(set_to_none=True)
    xs = [x.detach().clone().requires_grad_(True) for x in xs]
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
        y = blk.forward(*xs)
    y.float().backward(cot)
    return y.detach().float(), [x.grad for x in xs], {k: p.grad.detach().clone() for k, p in blk.named_parameters()}


class NNUNet(nn.Module):
    """Any depth / width plan of ``unet_spec`` (the reference's UNet, UNet_S, UNet_T and BASELINE config 4's depth-5 net)."""

    def __init__(self, n_channels: int, n_classes: int, bilinear: bool,
                 widths: Sequence[int] = (64, 128, 256, 512, 1024)):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        self.depth = len(widths) - 1
        for name, kind, args in unet_spec(n_channels, n_classes, bilinear, widths):
            blk = nn.Module()
            if kind == "double_conv":
                blk = _conv_bn_relu_twice(*args)
            elif kind == "down":                                   # unet_parts.py:31-34
                blk.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _conv_bn_relu_twice(*args))
            elif kind == "up":                                     # unet_parts.py:69-74
                cin, cout, mid = args
                blk.up = (nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) if bilinear
                          else nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2))
                blk.conv = _conv_bn_relu_twice(cin, cout, mid)
            else:                                                  # unet_parts.py:103
                blk.conv = nn.Conv2d(args[0], args[1], kernel_size=1)
            setattr(self, name, blk)

    def forward(self, x):
        skips = [self.inc.double_conv(x)]
        for k in range(1, self.depth + 1):
            d = getattr(self, f"down{k}").maxpool_conv
            skips.append(d[1].double_conv(d[0](skips[-1])))
        y = skips[-1]
        for j in range(1, self.depth + 1):
            blk, skip = getattr(self, f"up{j}"), skips[self.depth - j]
            y = blk.up(y)
            dy, dx = skip.shape[2] - y.shape[2], skip.shape[3] - y.shape[3]
            y = F.pad(y, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])          # unet_parts.py:85-88
            y = blk.conv.double_conv(torch.cat([skip, y], dim=1))                  # unet_parts.py:95 (skip first)
        return self.outc.conv(y)


class NNStepper:
    """train.py:80-85 (optimizer, criterion) + one call per optimizer step (train.py:113-159)."""

    def __init__(self, model: NNUNet, lr: float = 1e-5, amp: bool = False, weight_decay: float = 1e-8,
                 momentum: float = 0.999, gradient_clipping: float = 1.0, boundary_weight_multiclass: float = 0.0):
        self.model, self.amp, self.clip, self.bmc = model, amp, gradient_clipping, boundary_weight_multiclass
        self.optimizer = torch.optim.RMSprop(model.parameters(), lr=lr, weight_decay=weight_decay, momentum=momentum,
                                             foreach=True)
        self.criterion = nn.CrossEntropyLoss() if model.n_classes > 1 else nn.BCEWithLogitsLoss()

    def step(self, images: torch.Tensor, masks: torch.Tensor) -> Dict[str, torch.Tensor]:
        model = self.model
        model.train()
        true_masks = masks.clone()
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=self.amp):
            pred = model(images)
            if model.n_classes == 1:
                true_masks //= 2
                t = true_masks.float()
                first = self.criterion(pred.squeeze(1), t)
                dice = L.dice_loss(torch.sigmoid(pred.squeeze(1)), t, multiclass=False)
                boundary = L.boundary_loss(pred.squeeze(1), t, edge_width=51, edge_weight=15)
                loss = first + dice + 0.25 * boundary
            else:
                first = self.criterion(pred, true_masks)
                dice = L.dice_loss(F.softmax(pred, dim=1).float(),
                                   F.one_hot(true_masks, model.n_classes).permute(0, 3, 1, 2).float(), multiclass=True)
                boundary = torch.zeros(())
                loss = first + dice
                if self.bmc:
                    boundary = L.boundary_loss(pred, true_masks.float(), edge_width=51, edge_weight=7)
                    loss = loss + self.bmc * boundary
        if torch.isnan(loss).any():
            raise RuntimeError("Fatal: NaN loss detected!")
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), self.clip)
        info = {"bce" if model.n_classes == 1 else "ce": first.detach().float(), "dice": dice.detach().float(),
                "boundary": boundary.detach().float(), "loss": loss.detach().float(), "grad_norm": total_norm.detach(),
                "logits": pred.detach().float(),
                "grads": {k: p.grad.detach().clone() for k, p in model.named_parameters()}}      # after clipping
        self.optimizer.step()
        return info
