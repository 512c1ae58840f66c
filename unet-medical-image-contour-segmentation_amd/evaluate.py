"""Validation metric of /root/reference/evaluate.py:12-171 (raw Dice; the OpenCV post-processing
branch and PNG dumps are host-side and outside the hot-path scope)."""
from __future__ import annotations

import torch

from . import ops
from .utils.dice_score import dice_coeff


@torch.inference_mode()
def evaluate(net, dataloader, device, amp, epoch_pred_dir=None, postprocess=False):
    if postprocess:
        raise NotImplementedError("OpenCV post-processing (utils/post_process.py) is out of scope; pass postprocess=False")
    net.eval()
    batches = list(dataloader)
    num_val_batches = len(batches)
    dice_score = 0
    min_dice = 10
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        for batch in batches:
            image, mask_true = batch["image"], batch["mask"]
            image = image.to(device=device, dtype=torch.float32, memory_format=torch.channels_last)
            mask_true = mask_true.to(device=device, dtype=torch.float32)
            mask_pred = net(image)
            if net.n_classes == 1:
                mask_true = torch.div(mask_true, 2, rounding_mode="floor")                      # evaluate.py:56
                assert mask_true.min() >= 0 and mask_true.max() <= 1, "True mask indices should be in [0, 1]"
                pred = ops.threshold_mask(mask_pred.squeeze(1))    # sigmoid(x) > 0.5  <=>  x > 0   (evaluate.py:60-62)
                d = dice_coeff(pred, mask_true, reduce_batch_first=False)
            else:
                idx = ops.argmax_classes(mask_pred)                                             # evaluate.py:111
                d = dice_coeff((idx == 2).float(), (mask_true == 2).float(), reduce_batch_first=False)
            dice_score += d
            if d < min_dice:
                min_dice = d
    net.train()
    n = max(num_val_batches, 1)
    return dice_score / n, dice_score / n, min_dice
