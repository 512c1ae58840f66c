"""Validation metric of /root/reference/evaluate.py:12-171: eval-mode forward (running statistics folded into the conv
epilogue), device-side masks, raw Dice and -- with postprocess=True -- the Dice after utils/post_process.postprocess_mask,
which here runs on the device for the whole batch (the reference copies every image to the host for OpenCV).  PNG dumps
(epoch_pred_dir) are host-side file output and are not written."""
from __future__ import annotations

import torch

from . import ops
from .utils.dice_score import dice_coeff
from .utils.post_process import postprocess_mask


@torch.inference_mode()
def evaluate(net, dataloader, device, amp, epoch_pred_dir=None, postprocess=True, process_group=None):
    """evaluate.py:12-171.  `postprocess=True` is the reference's default (evaluate.py:13): the second return value is then
    the Dice after post-processing and the minimum is taken over min(raw, post-processed) per batch (evaluate.py:85).
    Batches are consumed lazily; under torch.distributed (every rank evaluating its shard of the validation set) the Dice
    sums and batch counts are all-reduced so that every rank returns the metric of the whole set."""
    net.eval()
    num_val_batches = 0
    dice_score = torch.zeros((), dtype=torch.float32, device=device)
    dice_post = torch.zeros((), dtype=torch.float32, device=device)
    min_dice = torch.full((), 10.0, dtype=torch.float32, device=device)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        for batch in dataloader:
            num_val_batches += 1
            image, mask_true = batch["image"], batch["mask"]
            image = image.to(device=device, dtype=torch.float32, memory_format=torch.channels_last)
            mask_true = mask_true.to(device=device, dtype=torch.float32)
            mask_pred = net(image)
            if net.n_classes == 1:
                mask_true = torch.div(mask_true, 2, rounding_mode="floor")                      # evaluate.py:56
                assert mask_true.min() >= 0 and mask_true.max() <= 1, "True mask indices should be in [0, 1]"
                pred = ops.threshold_mask(mask_pred.squeeze(1))    # sigmoid(x) > 0.5  <=>  x > 0   (evaluate.py:60-62)
                d = dice_coeff(pred, mask_true, reduce_batch_first=False)
                cur = d
                if postprocess:
                    # evaluate.py:71-78 literally: the binary mask goes in coded {0,255}, postprocess_mask looks for class 2
                    # (finds none) and `processed // 255` is the prediction that is scored
                    coded = (pred * 255).to(torch.uint8)
                    processed = (postprocess_mask(coded) // 255).float()
                    dp = dice_coeff(processed, mask_true, reduce_batch_first=False)
                    dice_post += dp
                    cur = torch.minimum(d, dp)                                                  # evaluate.py:85
            else:
                idx = ops.argmax_classes(mask_pred)                                             # evaluate.py:111
                true_c = (mask_true == 2).float()
                d = dice_coeff((idx == 2).float(), true_c, reduce_batch_first=False)
                cur = d
                if postprocess:
                    processed = postprocess_mask(idx.to(torch.uint8))                           # evaluate.py:125-134
                    dice_post += dice_coeff((processed == 2).float(), true_c, reduce_batch_first=False)
            dice_score += d
            min_dice = torch.minimum(min_dice, cur.float())
    net.train()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        acc = torch.stack([dice_score, dice_post, torch.tensor(float(num_val_batches), device=device)])
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=process_group)
        dist.all_reduce(min_dice, op=dist.ReduceOp.MIN, group=process_group)
        dice_score, dice_post, n = acc[0], acc[1], acc[2].clamp_min(1.0)
    else:
        n = max(num_val_batches, 1)
    if not postprocess:
        dice_post = dice_score                                                                  # evaluate.py:168-169
    return dice_score / n, dice_post / n, min_dice
