"""Oracle (test infrastructure): one optimizer step of /root/reference/train.py:113-159.

The reference's train.py cannot be imported (it needs cv2 and two modules that are not
in the tree), so the statement sequence is restated here:

    masks_pred = model(images)                                   train.py:117
    true_masks //= 2 ; BCE + dice_loss + 0.25*boundary_loss      train.py:118-134   (n_classes == 1)
    CE + multiclass dice_loss                                    train.py:136-142   (n_classes  > 1)
    NaN check                                                    train.py:149-151
    zero_grad -> backward -> clip_grad_norm_(1.0) -> RMSprop     train.py:153-159

clip_grad_norm_ and RMSprop(lr, alpha=0.99, eps=1e-8, weight_decay=1e-8, momentum=0.999,
centered=False) (train.py:80-81) are written out arithmetically so that the fused HIP
optimizer kernel has an independent statement to be compared with.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import losses_ref as L
from . import unet_ref as U


def seg_loss(logits: torch.Tensor, masks: torch.Tensor, n_classes: int,
             boundary_weight_multiclass: float = 0.0) -> Dict[str, torch.Tensor]:
    """Loss assembly of train.py:118-142.  ``masks`` is int64 in {0,1,2,..}."""
    out: Dict[str, torch.Tensor] = {}
    if n_classes == 1:
        t = (masks // 2).float()                                   # train.py:119
        l = logits.squeeze(1)
        out["bce"] = L.bce_with_logits_mean(l, t)                  # train.py:121
        out["dice"] = L.dice_loss(torch.sigmoid(l), t, multiclass=False)   # train.py:123
        out["boundary"] = L.boundary_loss(l, t, edge_width=51, edge_weight=15)  # train.py:134
        out["loss"] = out["bce"] + out["dice"] + 0.25 * out["boundary"]
    else:
        out["ce"] = L.cross_entropy_mean(logits, masks)            # train.py:137
        oh = F.one_hot(masks, n_classes).permute(0, 3, 1, 2).float()
        out["dice"] = L.dice_loss(F.softmax(logits, dim=1).float(), oh, multiclass=True)  # :138-142
        out["loss"] = out["ce"] + out["dice"]
        if boundary_weight_multiclass:
            # the commented-out branch train.py:143-147 (cfg-4 asks for it): 4-D path, channel 1
            out["boundary"] = L.boundary_loss(logits, masks.float(), edge_width=51, edge_weight=7)
            out["loss"] = out["loss"] + boundary_weight_multiclass * out["boundary"]
    return out


def clip_coef(total_norm: torch.Tensor, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6))."""
    return torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)


def rmsprop_update(p, g, sq, buf, lr, alpha=0.99, eps=1e-8, weight_decay=1e-8, momentum=0.999):
    """torch.optim.RMSprop single-tensor rule (non-centered, momentum>0)."""
    g = g + weight_decay * p
    sq = alpha * sq + (1 - alpha) * g * g
    avg = sq.sqrt() + eps
    buf = momentum * buf + g / avg
    p = p - lr * buf
    return p, sq, buf


def train_step(state: U.State, opt: Optional[Dict[str, Dict[str, torch.Tensor]]],
               images: torch.Tensor, masks: torch.Tensor, *, n_classes: int, bilinear: bool,
               depth: int = 4, lr: float = 1e-5, weight_decay: float = 1e-8,
               momentum: float = 0.999, gradient_clipping: float = 1.0,
               boundary_weight_multiclass: float = 0.0, amp: bool = False):
    """Returns (new_state, new_opt, info).  Pure: inputs are not mutated.
    amp=True wraps forward + loss in torch.autocast('cpu', bfloat16) as train.py:116 does on a CUDA-less host (the
    reference CLI's default, train.py:233; GradScaler is auto-disabled there).  The bf16 leg is a TIMING baseline
    (bench.py cpu_baseline): the golden fixtures pin the fp32 leg only."""
    keys = U.param_keys(state)
    work = {k: (v.detach().clone().requires_grad_(True) if k in keys else v.detach().clone())
            for k, v in state.items()}
    new_buffers: U.State = {}
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
        logits = U.unet_forward(images, work, bilinear, depth, training=True, new_buffers=new_buffers)
        terms = seg_loss(logits, masks, n_classes, boundary_weight_multiclass)
    loss = terms["loss"]
    if torch.isnan(loss).any():
        raise RuntimeError("Fatal: NaN loss detected!")           # train.py:151
    grads = torch.autograd.grad(loss, [work[k] for k in keys])
    total_norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = clip_coef(total_norm, gradient_clipping)
    if opt is None:
        opt = {k: {"square_avg": torch.zeros_like(state[k]), "momentum_buffer": torch.zeros_like(state[k])}
               for k in keys}
    new_state = {k: v.detach().clone() for k, v in state.items()}
    new_state.update(new_buffers)
    new_opt = {}
    for k, g in zip(keys, grads):
        p, sq, buf = rmsprop_update(state[k], g * coef, opt[k]["square_avg"], opt[k]["momentum_buffer"],
                                    lr, weight_decay=weight_decay, momentum=momentum)
        new_state[k] = p
        new_opt[k] = {"square_avg": sq, "momentum_buffer": buf}
    info = {k: v.detach() for k, v in terms.items()}
    info["logits"] = logits.detach()
    info["grad_norm"] = total_norm
    info["grads"] = {k: g for k, g in zip(keys, grads)}
    return new_state, new_opt, info


def evaluate_dice(state: U.State, images, masks, *, n_classes: int, bilinear: bool, depth: int = 4):
    """Metric of /root/reference/evaluate.py:43-66,109-123 (raw Dice, no post-process)."""
    with torch.no_grad():
        logits = U.unet_forward(images, state, bilinear, depth, training=False)
        if n_classes == 1:
            t = (masks // 2).float()
            pred = (torch.sigmoid(logits.squeeze(1)) > 0.5).float()
            return L.dice_coeff(pred, t, reduce_batch_first=False), logits
        idx = logits.argmax(dim=1)
        return L.dice_coeff((idx == 2).float(), (masks == 2).float(), reduce_batch_first=False), logits
