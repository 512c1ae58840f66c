"""Oracle (test infrastructure): functional fp32 restatement of the UNet blocks.

Follows /root/reference/unet/unet_parts.py:7-106 and
/root/reference/unet/unet_model.py:8-38.  State is a flat dict keyed exactly like
the reference's ``state_dict()``, so a reference
checkpoint drives these functions unchanged.  All maths are stock PyTorch CPU
ops; gradients come from torch autograd over these functions.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

State = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- specs
def unet_spec(n_channels: int, n_classes: int, bilinear: bool,
              widths: Sequence[int] = (64, 128, 256, 512, 1024)):
    """Channel plan of unet_model.py:15-25 generalised to any depth/base width.

    Returns a list of (name, kind, args) tuples in forward order.
    """
    widths = list(widths)
    depth = len(widths) - 1
    factor = 2 if bilinear else 1
    spec = [("inc", "double_conv", (n_channels, widths[0], widths[0]))]
    for k in range(1, depth + 1):
        cout = widths[k] // factor if k == depth else widths[k]
        spec.append((f"down{k}", "down", (widths[k - 1], cout, cout)))
    for j in range(1, depth + 1):
        cin = widths[depth - j + 1]
        cout = widths[depth - j] // factor if j < depth else widths[0]
        mid = cin // 2 if bilinear else cout
        spec.append((f"up{j}", "up", (cin, cout, mid)))
    spec.append(("outc", "outconv", (widths[0], n_classes)))
    return spec


def init_state(n_channels: int, n_classes: int, bilinear: bool,
               widths: Sequence[int] = (64, 128, 256, 512, 1024),
               seed: int = 0) -> State:
    """Random-init state with the reference's key names and PyTorch default inits
    (kaiming_uniform(a=sqrt(5)) for conv weights, U(-1/sqrt(fan_in), ..) for bias)."""
    g = torch.Generator().manual_seed(seed)
    st: State = {}

    def conv_w(co, ci, k):
        fan_in = ci * k * k
        bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform with a=sqrt(5)
        return (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * bound

    def bn(prefix, c):
        st[prefix + ".weight"] = torch.ones(c)
        st[prefix + ".bias"] = torch.zeros(c)
        st[prefix + ".running_mean"] = torch.zeros(c)
        st[prefix + ".running_var"] = torch.ones(c)
        st[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    def dconv(prefix, cin, cout, mid):
        st[prefix + ".double_conv.0.weight"] = conv_w(mid, cin, 3)
        bn(prefix + ".double_conv.1", mid)
        st[prefix + ".double_conv.3.weight"] = conv_w(cout, mid, 3)
        bn(prefix + ".double_conv.4", cout)

    for name, kind, args in unet_spec(n_channels, n_classes, bilinear, widths):
        if kind == "double_conv":
            dconv(name, *args)
        elif kind == "down":
            dconv(name + ".maxpool_conv.1", *args)
        elif kind == "up":
            cin, cout, mid = args
            if not bilinear:
                # ConvTranspose2d weight layout is [in, out, kh, kw]; torch computes
                # fan_in from dim 1 (= out channels) * k*k for this layout.
                fan_in = (cin // 2) * 4
                bound = 1.0 / math.sqrt(fan_in)
                st[name + ".up.weight"] = (torch.rand(cin, cin // 2, 2, 2, generator=g) * 2 - 1) * bound
                st[name + ".up.bias"] = (torch.rand(cin // 2, generator=g) * 2 - 1) * bound
            dconv(name + ".conv", cin, cout, mid)
        else:
            cin, ncls = args
            bound = 1.0 / math.sqrt(cin)
            st[name + ".conv.weight"] = (torch.rand(ncls, cin, 1, 1, generator=g) * 2 - 1) * bound
            st[name + ".conv.bias"] = (torch.rand(ncls, generator=g) * 2 - 1) * bound
    return st


# --------------------------------------------------------------------------- blocks
class _BatchNormTrain(torch.autograd.Function):
    """Train-mode BatchNorm2d with the closed-form backward the HIP kernels implement
    (SURVEY.md A.3): dgamma = sum(dz*xhat), dbeta = sum(dz),
    dx = gamma*rstd*(dz - dbeta/n - xhat*dgamma/n).  Written out (instead of autograd through
    mean/var) because that composition loses ~3 digits in fp32 to cancellation."""

    @staticmethod
    def forward(ctx, x, w, b):
        dims = (0, 2, 3)
        mean = x.mean(dims)
        var = ((x - mean[None, :, None, None]) ** 2).mean(dims)      # biased
        rstd = torch.rsqrt(var + BN_EPS)
        xhat = (x - mean[None, :, None, None]) * rstd[None, :, None, None]
        ctx.save_for_backward(xhat, w, rstd)
        ctx.mark_non_differentiable(mean, var)
        return xhat * w[None, :, None, None] + b[None, :, None, None], mean, var

    @staticmethod
    def backward(ctx, dz, _dm, _dv):
        xhat, w, rstd = ctx.saved_tensors
        dims = (0, 2, 3)
        n = dz.numel() // dz.shape[1]
        dbeta = dz.sum(dims)
        dgamma = (dz * xhat).sum(dims)
        dx = (w * rstd)[None, :, None, None] * (dz - (dbeta / n)[None, :, None, None]
                                                 - xhat * (dgamma / n)[None, :, None, None])
        return dx, dgamma, dbeta


def _bn_relu(x, st: State, prefix: str, training: bool, new_buffers: State | None):
    """BatchNorm2d (unet_parts.py:16,19) + ReLU (unet_parts.py:17,20).

    Train mode: batch mean / biased var normalise; running stats take momentum 0.1
    with the *unbiased* variance.  Running-stat updates are returned
    through ``new_buffers`` instead of mutating ``st``.
    """
    w, b = st[prefix + ".weight"], st[prefix + ".bias"]
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    if training:
        n = x.numel() // x.shape[1]
        y, mean, var = _BatchNormTrain.apply(x, w, b)
        if new_buffers is not None:
            with torch.no_grad():
                new_buffers[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                new_buffers[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * (n / max(n - 1, 1))
                new_buffers[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
    else:
        scale = w * torch.rsqrt(rv + BN_EPS)
        y = x * scale[None, :, None, None] + (b - rm * scale)[None, :, None, None]
    return F.relu(y)


def double_conv(x, st: State, prefix: str, training=True, new_buffers=None):
    """unet_parts.py:7-24: (conv3x3 pad1 no-bias -> BN -> ReLU) x 2."""
    x = F.conv2d(x, st[prefix + ".double_conv.0.weight"], None, padding=1)
    x = _bn_relu(x, st, prefix + ".double_conv.1", training, new_buffers)
    x = F.conv2d(x, st[prefix + ".double_conv.3.weight"], None, padding=1)
    x = _bn_relu(x, st, prefix + ".double_conv.4", training, new_buffers)
    return x


def down(x, st: State, prefix: str, training=True, new_buffers=None):
    """unet_parts.py:26-37: MaxPool2d(2) then DoubleConv."""
    x = F.max_pool2d(x, 2)
    return double_conv(x, st, prefix + ".maxpool_conv.1", training, new_buffers)


def up(x1, x2, st: State, prefix: str, bilinear: bool, training=True, new_buffers=None):
    """unet_parts.py:62-98: upsample x1, zero-pad to x2's size, cat([x2, x1]), DoubleConv."""
    if bilinear:
        x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        x1 = F.conv_transpose2d(x1, st[prefix + ".up.weight"], st[prefix + ".up.bias"], stride=2)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    x = torch.cat([x2, x1], dim=1)
    return double_conv(x, st, prefix + ".conv", training, new_buffers)


def out_conv(x, st: State, prefix: str):
    """unet_parts.py:100-106: conv1x1 with bias."""
    return F.conv2d(x, st[prefix + ".conv.weight"], st[prefix + ".conv.bias"])


def unet_forward(x, st: State, bilinear: bool, depth: int = 4, training=True, new_buffers=None):
    """unet_model.py:27-38 wiring for any depth."""
    feats = [double_conv(x, st, "inc", training, new_buffers)]
    for k in range(1, depth + 1):
        feats.append(down(feats[-1], st, f"down{k}", training, new_buffers))
    y = feats[-1]
    for j in range(1, depth + 1):
        y = up(y, feats[depth - j], st, f"up{j}", bilinear, training, new_buffers)
    return out_conv(y, st, "outc")


def param_keys(st: State) -> List[str]:
    """Keys that are nn.Parameters in the reference (everything except BN buffers)."""
    return [k for k in st if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]
