"""Dice metric / loss on the HIP reduction kernels.

Drop-in surface of /root/reference/utils/dice_score.py:5-36 (same names, argument meaning, the two
assertions, the `sets_sum == 0` branch).  One pass over the inputs produces the three sums
{sum x*t, sum x, sum t} per group (wave-shuffle reductions, csrc/loss.hip); the ratio and the mean
over groups are formed on device.
"""
from __future__ import annotations

import torch
from torch import Tensor

from .. import ops


def dice_coeff(input: Tensor, target: Tensor, reduce_batch_first: bool = False, epsilon: float = 1e-6):
    assert input.size() == target.size()
    assert input.dim() == 3 or not reduce_batch_first
    if input.dim() == 2 or reduce_batch_first:
        ngroups = 1                                   # sums over every dimension
    else:
        ngroups = 1
        for d in input.shape[:-2]:
            ngroups *= int(d)                         # per-image ratios, then the mean
    group_len = input.numel() // ngroups
    return ops.DiceCoeffFn.apply(input, target, ngroups, group_len, float(epsilon))


def multiclass_dice_coeff(input: Tensor, target: Tensor, reduce_batch_first: bool = False, epsilon: float = 1e-6):
    return dice_coeff(input.flatten(0, 1), target.flatten(0, 1), reduce_batch_first, epsilon)


def dice_loss(input: Tensor, target: Tensor, multiclass: bool = False):
    fn = multiclass_dice_coeff if multiclass else dice_coeff
    return 1 - fn(input, target, reduce_batch_first=True)
