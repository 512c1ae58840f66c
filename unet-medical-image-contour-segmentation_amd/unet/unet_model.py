"""UNet assemblies on the HIP blocks.

Drop-in surface of /root/reference/unet/unet_model.py: `UNet`, `UNet_S`, `UNet_T`
(n_channels, n_classes, bilinear=False), attributes n_channels / n_classes / bilinear, children
inc, down1..downN, up1..upN, outc, forward(x[B,C,H,W]) -> logits[B,n_classes,H,W]
(unet_model.py:8-38, 52-82, 96-126).  `UNetDepth` builds the same wiring for any width list
(BASELINE config 4 is the 5-level variant).  UNet_SA / use_checkpointing are out of scope
(attention is used by no config; use_checkpointing is broken upstream).
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from .. import ops
from .unet_parts import DoubleConv, Down, OutConv, Up


class UNetDepth(nn.Module):
    def __init__(self, n_channels, n_classes, bilinear=False, widths: Sequence[int] = (64, 128, 256, 512, 1024)):
        super().__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear
        self.widths = tuple(int(w) for w in widths)
        self.depth = len(self.widths) - 1
        shrink = 2 if bilinear else 1
        w = self.widths
        self.inc = DoubleConv(n_channels, w[0])
        for k in range(1, self.depth + 1):
            cout = w[k] // shrink if k == self.depth else w[k]
            setattr(self, f"down{k}", Down(w[k - 1], cout))
        for j in range(1, self.depth + 1):
            cin = w[self.depth - j + 1]
            cout = w[self.depth - j] // shrink if j < self.depth else w[0]
            setattr(self, f"up{j}", Up(cin, cout, bilinear))
        self.outc = OutConv(w[0], n_classes)
        # activation dtype of the HIP path when no autocast context is active
        self.compute_dtype = torch.float32

    def forward_nhwc(self, x):
        # Every encoder DoubleConv hands its output to the max-pool of the next Down AND to the skip connection, the last
        # decoder DoubleConv hands its output to OutConv: the blocks are told their consumer so that BatchNorm + ReLU and
        # the consumer run as one kernel where csrc/bn_fused.hip covers the shape (unet_model.py:28-38 is the same graph).
        skips = []
        skip, pooled = self.inc.nhwc(x, tail="pool")
        skips.append(skip)
        for k in range(1, self.depth + 1):
            block = getattr(self, f"down{k}").maxpool_conv[1]
            if k < self.depth:
                skip, pooled = block.nhwc(pooled, tail="pool")
                skips.append(skip)
            else:
                h = block.nhwc(pooled, tail=self._up_tail(1, skips))
        for j in range(1, self.depth + 1):
            up = getattr(self, f"up{j}")
            tail = self.outc if j == self.depth else self._up_tail(j + 1, skips)
            h = up.nhwc(h, skips[self.depth - j], tail=tail, upsampled=up.bilinear)
        return h

    def _up_tail(self, j: int, skips):
        """What consumes the tensor handed to Up block j: with bilinear up-sampling its nn.Upsample + F.pad (unet_parts.py:80,85-88)
        are the ONLY reader, so the producer block is told the skip's extent and returns the up-sampled tensor itself -- one
        kernel with its last BatchNorm + ReLU in training (csrc/pool_up.hip), the separate up-sampling kernel otherwise."""
        if not getattr(self, f"up{j}").bilinear:
            return None
        s = skips[self.depth - j]
        return ("up", int(s.shape[1]), int(s.shape[2]))

    def forward(self, x):
        dt = ops.compute_dtype(self.compute_dtype)
        return ops.to_nchw(self.forward_nhwc(ops.to_nhwc(x, dt)))

    def use_checkpointing(self):
        raise NotImplementedError("use_checkpointing is broken in the reference (unet_model.py:40-50) and out of scope")


class UNet(UNetDepth):
    """64-128-256-512-1024 (unet_model.py:15-25)."""

    def __init__(self, n_channels, n_classes, bilinear=False):
        super().__init__(n_channels, n_classes, bilinear, (64, 128, 256, 512, 1024))


class UNet_S(UNetDepth):
    """16-32-64-128-256 (unet_model.py:103-113)."""

    def __init__(self, n_channels, n_classes, bilinear=False):
        super().__init__(n_channels, n_classes, bilinear, (16, 32, 64, 128, 256))


class UNet_T(UNetDepth):
    """8-16-32-64-128 (unet_model.py:59-69)."""

    def __init__(self, n_channels, n_classes, bilinear=False):
        super().__init__(n_channels, n_classes, bilinear, (8, 16, 32, 64, 128))
