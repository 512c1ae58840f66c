"""UNet building blocks on the MI355X HIP kernels.

Drop-in surface of /root/reference/unet/unet_parts.py (constructor signatures, forward signatures,
sub-module attribute names and therefore state_dict keys -- SURVEY.md A.1):

    DoubleConv(in_channels, out_channels, mid_channels=None)      unet_parts.py:7-24
    Down(in_channels, out_channels)                               unet_parts.py:26-37
    Up(in_channels, out_channels, bilinear=True)                  unet_parts.py:62-98
    OutConv(in_channels, out_channels)                            unet_parts.py:100-106

The nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d children are PARAMETER CONTAINERS only (same
init, same keys, visible to optimizers / state_dict / .to()); their torch forward is never called.
Each block's forward enqueues the hand-written kernels of csrc/ through ops.py.  Public forward
takes and returns logical NCHW tensors like the reference; `nhwc()` methods are the internal
channels-last entry points UNet.forward chains without layout round trips.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


CPAD = 64          # channel granularity of the MFMA conv kernels (64-channel slabs, 64-byte K chunks)


def _rup(c: int) -> int:
    return (c + CPAD - 1) // CPAD * CPAD


def _pad_c(t, cpad: int):
    """Zero-pad the channel (last) axis of an NHWC tensor up to `cpad` (differentiable torch plumbing)."""
    c = t.shape[-1]
    return t if c == cpad else torch.nn.functional.pad(t, (0, cpad - c))


def _is_up(tail) -> bool:
    return isinstance(tail, tuple) and len(tail) == 3 and tail[0] == "up"


def _finish_tail(z, tail):
    """The consumer of a DoubleConv's output as separate kernels: None -> z; "pool" -> (z as skip, maxpool2(z));
    ("up", Ho, Wo) -> bilinear x2 + zero padding to Ho x Wo (the next Up block's nn.Upsample + F.pad); an OutConv -> its logits."""
    if tail is None:
        return z
    if tail == "pool":
        return ops.PoolSplitFn.apply(z)
    if _is_up(tail):
        return ops.UpsampleBilinearPadFn.apply(z, tail[1], tail[2])
    return tail.nhwc(z)


def _conv_bn_relu(x0, x1, conv: nn.Conv2d, bn: nn.BatchNorm2d, training: bool, keep_padded: bool = False,
                  x0_channels: int = None, tail=None, defer: bool = False, pre_coef=None, bnsum_pub=None, bnsum_use=None):
    """One (conv3x3 -> BatchNorm -> ReLU) layer, plus its consumer `tail` (see _finish_tail) -- fused into the BatchNorm
    kernels where csrc/bn_fused.hip covers the shape (training, 64-aligned layers), separate kernels otherwise.
    Layers whose channel counts are not multiples of 64 (the small-width UNet_S / UNet_T of unet_model.py:52-126) are computed as the next larger 64-aligned layer with zero filters / unit
    gamma in the padding, so that they use the same MFMA kernels as the full-width UNet instead of the generic scalar
    kernels.  Default (ops.NARROW_IO): the tensors themselves stay at their real channel count (ConvBnReluNarrowFn).
    Otherwise / for channel counts that are not 16-byte multiples: zero-padded 64-channel tensors -- padded input
    channels meet zero filter taps, padded output channels are exactly 0 before and after BatchNorm+ReLU (mean 0,
    shift 0), and their gradients never reach a parameter (the slices below drop them).
    `x0_channels`: x0 is already such a padded tensor and only its first x0_channels channels are real.
    `keep_padded`: return the padded tensor (DoubleConv hands it to its second conv without a copy)."""
    momentum = 0.1 if bn.momentum is None else bn.momentum
    w = conv.weight
    Cout, Cin = w.shape[0], w.shape[1]
    C0 = x0_channels if x0_channels is not None else x0.shape[-1]
    C1 = 0 if x1 is None else x1.shape[-1]
    stem = Cin <= 4 and x1 is None and x0_channels is None      # the stem kernels take any Cin <= 4 but want 64 outputs
    Cp0 = C0 if stem else (x0.shape[-1] if x0_channels is not None else _rup(C0))
    Cp1 = _rup(C1) if C1 else 0
    Cop = _rup(Cout)
    if Cp0 == C0 and Cp1 == C1 and Cop == Cout and x0_channels is None:
        if stem and training and tail is None and not defer and pre_coef is None and ops.stem_recompute_ok(x0, Cin, Cout):
            # the network's first layer: its conv output is recomputed by every consumer instead of stored
            return ops.StemConvBnReluFn.apply(x0, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                              momentum, bn.eps)
        args = (x0, x1, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, training, momentum,
                bn.eps)
        if defer:      # (the caller has checked ops.pre_fuse_ok: this layer's BatchNorm + ReLU is applied by its consumer)
            return ops.ConvBnReluFn.apply(*args, ops.TAIL_NONE, None, None, True, None)
        link = (bnsum_pub, bnsum_use) if training else (None, None)      # (ops.BnSumLink: see DoubleConv.nhwc)
        if training and tail == "pool" and ops.pool_tail_ok(x0, Cout):
            return ops.ConvBnReluFn.apply(*args, ops.TAIL_POOL, None, None, False, pre_coef, *link)
        if training and isinstance(tail, OutConv) and ops.head_tail_ok(x0, Cout, tail.conv.weight):
            return ops.ConvBnReluFn.apply(*args, ops.TAIL_HEAD, tail.conv.weight, tail.conv.bias, False, pre_coef, *link)
        if training and _is_up(tail) and ops.up_tail_ok(x0, Cout, tail[1], tail[2]):
            return ops.ConvBnReluFn.apply(*args, ops.TAIL_UP, None, None, False, pre_coef, *link, (tail[1], tail[2]))
        return _finish_tail(ops.ConvBnReluFn.apply(*args, ops.TAIL_NONE, None, None, False, pre_coef, *link), tail)
    if defer or pre_coef is not None:
        raise RuntimeError("a deferred BatchNorm+ReLU needs 64-aligned layers (ops.pre_fuse_ok)")
    if ops.NARROW_IO and x0_channels is None:
        # tensors keep their real channel count in HBM, only the arithmetic is padded (ops.ConvBnReluNarrowFn); the
        # 1- / 3-channel image is widened to one 16-byte piece so that it can be fetched like any other activation
        vec = 16 // x0.element_size()
        x0n = _pad_c(x0, (C0 + vec - 1) // vec * vec)
        if ops.narrow_ok(x0n, x1, Cout):
            return _finish_tail(ops.ConvBnReluNarrowFn.apply(x0n, x1, w, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                             bn.num_batches_tracked, training, momentum, bn.eps, C0), tail)
    x0p = x0 if x0_channels is not None else _pad_c(x0, Cp0)
    x1p = None if x1 is None else _pad_c(x1, Cp1)
    # filter [Cout, C0 + C1, 3, 3] -> [Cop, Cp0 + Cp1, 3, 3]: each source's channel block is padded separately
    wp = torch.nn.functional.pad(w[:, :C0], (0, 0, 0, 0, 0, Cp0 - C0))
    if C1:
        wp = torch.cat([wp, torch.nn.functional.pad(w[:, C0:], (0, 0, 0, 0, 0, Cp1 - C1))], dim=1)
    wp = torch.nn.functional.pad(wp, (0, 0, 0, 0, 0, 0, 0, Cop - Cout))
    gp = torch.nn.functional.pad(bn.weight, (0, Cop - Cout), value=1.0)
    bp = torch.nn.functional.pad(bn.bias, (0, Cop - Cout))
    rm = rv = None
    if bn.running_mean is not None:
        rm = torch.nn.functional.pad(bn.running_mean, (0, Cop - Cout))
        rv = torch.nn.functional.pad(bn.running_var, (0, Cop - Cout), value=1.0)
    zp = ops.ConvBnReluFn.apply(x0p, x1p, wp, gp, bp, rm, rv, bn.num_batches_tracked, training, momentum, bn.eps)
    if training and rm is not None:
        with torch.no_grad():
            bn.running_mean.copy_(rm[:Cout])
            bn.running_var.copy_(rv[:Cout])
    return _finish_tail(zp if keep_padded else zp[..., :Cout], tail)


class DoubleConv(nn.Module):
    """(conv3x3 -> BatchNorm -> ReLU) twice; the second input `x1` (optional) is the up-sampled half of
    the skip concatenation, read in place by the first conv (torch.cat is never materialised)."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        mid = mid_channels if mid_channels else out_channels
        layers = [
            nn.Conv2d(in_channels, mid, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        ]
        self.double_conv = nn.Sequential(*layers)

    def nhwc(self, x0, x1=None, tail=None):
        """`tail`: what consumes the block's output -- None (returns it), "pool" (returns (output, maxpool2(output)):
        unet_parts.py:32 of the next Down), ("up", Ho, Wo) (returns the output up-sampled x2 and zero-padded to Ho x Wo:
        unet_parts.py:80,85-88 of the next bilinear Up, which is then called with upsampled=True) or the network's OutConv
        (returns its logits)."""
        seq = self.double_conv
        mid = seq[0].weight.shape[0]
        w1, w2 = seq[0].weight, seq[3].weight
        cin1 = w1.shape[1]
        aligned1 = (cin1 <= 4 and x1 is None and mid == 64) or (x0.shape[-1] % CPAD == 0 and (x1 is None or x1.shape[-1] % CPAD == 0))
        if self.training and aligned1 and mid % CPAD == 0 and w2.shape[0] % CPAD == 0 and ops.pre_fuse_ok(x0, mid, w2.shape[0]):
            # the activation between the two convs never exists: the second conv's loaders apply the first layer's
            # BatchNorm + ReLU to its raw output on the way to the MFMAs
            y1, coef1 = _conv_bn_relu(x0, x1, seq[0], seq[1], True, defer=True)
            return _conv_bn_relu(y1, None, seq[3], seq[4], True, tail=tail, pre_coef=coef1)
        # the first layer's BatchNorm-backward sums are formed by the second conv's backward-data where the shapes allow
        # (ops.BnSumLink; bf16 training, 64-aligned layers -- the link stays unused everywhere else)
        link = ops.BnSumLink() if (self.training and ops.FUSE_BNSUM) else None
        h = _conv_bn_relu(x0, x1, seq[0], seq[1], self.training, keep_padded=True, bnsum_pub=link)
        return _conv_bn_relu(h, None, seq[3], seq[4], self.training, x0_channels=mid if h.shape[-1] != mid else None,
                             tail=tail, bnsum_use=link)

    def forward(self, x):
        return ops.to_nchw(self.nhwc(ops.to_nhwc(x, ops.compute_dtype(x.dtype if x.dtype == torch.bfloat16 else torch.float32))))


class Down(nn.Module):
    """MaxPool2d(2) then DoubleConv."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def nhwc(self, x):
        return self.maxpool_conv[1].nhwc(ops.MaxPool2Fn.apply(x))

    def nhwc_with_skip(self, x):
        """-> (x for the skip connection, block output); one backward pass sums both gradients of x."""
        skip, pooled = ops.PoolSplitFn.apply(x)
        return skip, self.maxpool_conv[1].nhwc(pooled)

    def forward(self, x):
        return ops.to_nchw(self.nhwc(ops.to_nhwc(x, ops.compute_dtype(x.dtype if x.dtype == torch.bfloat16 else torch.float32))))


class Up(nn.Module):
    """Upsample (bilinear x2, align_corners=True | ConvTranspose2d k2 s2), zero-pad to the skip's size,
    concat [skip, up] on channels, DoubleConv."""

    def __init__(self, in_channels, out_channels, bilinear=True, use_attention=False):
        super().__init__()
        if use_attention:
            raise NotImplementedError("SpatialAttention (UNet_SA only) is outside the hot-path scope (SURVEY.md section 2)")
        self.bilinear = bool(bilinear)
        if self.bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)
        self.use_attention = False
        self.attention = nn.Identity()

    def nhwc(self, x1, x2, tail=None, upsampled: bool = False):
        """`upsampled`: x1 already is the up-sampled, padded tensor (its producer ran with tail=("up", Ho, Wo))."""
        Ho, Wo = x2.shape[1], x2.shape[2]
        if upsampled:
            if not self.bilinear or tuple(x1.shape[1:3]) != (Ho, Wo):
                raise RuntimeError("Up.nhwc(upsampled=True) needs a bilinear block and an input of the skip's extent")
            u = x1
        elif self.bilinear:
            u = ops.UpsampleBilinearPadFn.apply(x1, Ho, Wo)
        else:
            u = ops.ConvTranspose2x2PadFn.apply(x1, self.up.weight, self.up.bias, Ho, Wo)
        return self.conv.nhwc(x2, u, tail=tail)          # channel order [skip, up] as unet_parts.py:95

    def forward(self, x1, x2):
        dt = ops.compute_dtype(x1.dtype if x1.dtype == torch.bfloat16 else torch.float32)
        return ops.to_nchw(self.nhwc(ops.to_nhwc(x1, dt), ops.to_nhwc(x2, dt)))


class OutConv(nn.Module):
    """1x1 convolution with bias; logits are returned in fp32."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def nhwc(self, x):
        return ops.OutConv1x1Fn.apply(x, self.conv.weight, self.conv.bias)

    def forward(self, x):
        return ops.to_nchw(self.nhwc(ops.to_nhwc(x, ops.compute_dtype(x.dtype if x.dtype == torch.bfloat16 else torch.float32))))
