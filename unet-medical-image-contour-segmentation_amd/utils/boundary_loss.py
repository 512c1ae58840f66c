"""boundary_loss on the HIP kernel (csrc/loss.hip: uh_boundary_loss).

Drop-in surface of /root/reference/utils/boundary_loss.py:5-45 with its literal behaviour
(SURVEY.md A.5): 4-D input -> channel 1 (C > 1) or squeeze; sigmoid only when min < -10 or
max > 10 (decided on device, no host sync); target == 255 is foreground; per region a 3-tap
dilation along the region's row-major gather order; (1 - IoU) + 0.5 * BCE; weighted combine.
The result is a 0-dim tensor WITHOUT gradient, exactly as in the reference (the threshold at
boundary_loss.py:101 cuts the graph).
"""
from __future__ import annotations

import torch

from .. import ops


def boundary_loss(pred_mask, target_mask, edge_width=64, edge_weight=5.0, smooth=1e-6):
    pred = pred_mask.detach()
    if pred.dim() == 4:
        pred = pred[:, 1, :, :] if pred.size(1) > 1 else pred.squeeze(1)
    if pred.dim() != 3:
        raise ValueError(f"boundary_loss expects [B,H,W] or [B,C,H,W] predictions, got {tuple(pred_mask.shape)}")
    if pred.dtype != torch.float32:
        pred = pred.float()
    B, H, W = pred.shape
    # one pixel stride inside an image is all the kernel needs: pred(b,h,w) = base + b*bs + (h*W+w)*ps
    ps = pred.stride(2) if W > 1 else (pred.stride(1) if H > 1 else 1)
    if (W > 1 and H > 1 and pred.stride(1) != W * ps) or ps <= 0:
        pred = pred.contiguous()
        ps = 1
    bs = pred.stride(0) if B > 1 else H * W * ps
    if bs <= 0:
        pred = pred.contiguous()
        ps, bs = 1, H * W
    target = target_mask.detach()
    if target.dtype != torch.float32:
        target = target.float()
    target = target.contiguous()
    if tuple(target.shape) != (B, H, W):
        raise ValueError(f"target {tuple(target.shape)} does not match predictions {(B, H, W)}")
    out = ops.boundary_loss_value(pred, ps, bs, target, B, H, W, int(edge_width), float(edge_weight), float(smooth))
    return out.view(())
