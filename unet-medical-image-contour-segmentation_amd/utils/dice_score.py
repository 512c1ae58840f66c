"""Dice metric / loss on the HIP reduction kernels.

Drop-in surface of /root/reference/utils/dice_score.py:5-36 (same names, argument meaning, the two
assertions, the `sets_sum == 0` branch).  One pass over the inputs produces the three sums
{sum x*t, sum x, sum t} per group (wave-shuffle reductions, csrc/loss.hip); the ratio and the mean
over groups are formed on device.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from .. import ops


def _groups(shape, all_dims: bool) -> int:
    """Number of independent Dice ratios: 1 when every dimension is summed, else one per leading index."""
    return 1 if all_dims else math.prod(int(d) for d in shape[:-2])


def dice_coeff(input: Tensor, target: Tensor, reduce_batch_first: bool = False, epsilon: float = 1e-6):
    assert input.size() == target.size()
    assert input.dim() == 3 or not reduce_batch_first
    ngroups = _groups(input.shape, all_dims=(input.dim() == 2 or reduce_batch_first))
    return ops.DiceCoeffFn.apply(input, target, ngroups, input.numel() // ngroups, float(epsilon))


def multiclass_dice_coeff(input: Tensor, target: Tensor, reduce_batch_first: bool = False, epsilon: float = 1e-6):
    # classes and images are folded into one leading axis: with reduce_batch_first this is ONE ratio over everything
    return dice_coeff(input.flatten(0, 1), target.flatten(0, 1), reduce_batch_first, epsilon)


def dice_loss(input: Tensor, target: Tensor, multiclass: bool = False):
    coeff = (multiclass_dice_coeff if multiclass else dice_coeff)(input, target, reduce_batch_first=True)
    return 1 - coeff
