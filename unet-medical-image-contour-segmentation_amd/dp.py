"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-process; data parallelism is introduced here at the
three places SURVEY.md 8(e) names:
  * gradients: the flat fp32 gradient buffer of FusedRMSprop is cut into contiguous buckets in
    backward-ready order (decoder first); a bucket is all-reduced (SUM, async, on RCCL's own stream) as
    soon as its last gradient has been produced, so the transfers overlap the encoder backward.  SUM, not
    mean: the loss of every rank is already normalised by the GLOBAL batch (BCE mean over n*world pixels,
    Dice ratio of all-reduced sums), so the rank gradients are the per-shard terms of ONE global gradient;
  * Dice / BCE: the 4 (binary) or 1+3C (multi-class) partial sums are all-reduced before the
    ratio is formed, so loss value and gradient are those of the GLOBAL batch;
  * BatchNorm: per-rank batch statistics (what stock DDP does) -- stated in DESIGN.md.
Works on any backend (tests use gloo on CPU tensors for the bucket logic).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def make_sum_reducer(group=None):
    """Callable that all-reduces (sum) a small tensor in place; None when not distributed."""
    if world_size(group) == 1:
        return None

    def _reduce(t: torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    return _reduce


class BucketedGradSync:
    """All-reduce (sum; `average=True` for a mean) a flat gradient buffer in contiguous buckets as their slices become
    ready."""

    def __init__(self, flat_grad: torch.Tensor, slices: Sequence[Tuple[int, int]], bucket_bytes: int = 8 << 20,
                 group=None, average: bool = False):
        self.flat = flat_grad
        self.group = group
        self.world = world_size(group)
        self.slices = list(slices)                  # (offset, numel) per parameter, in flat order
        limit = max(1, bucket_bytes // flat_grad.element_size())
        self.buckets: List[Tuple[int, int]] = []    # (start, end) element ranges
        self.bucket_of: List[int] = []
        start = 0
        count = 0
        for off, n in self.slices:
            if count and count + n > limit:
                self.buckets.append((start, off))
                start, count = off, 0
            self.bucket_of.append(len(self.buckets))
            count += n
        self.buckets.append((start, flat_grad.numel()))
        self.members = [0] * len(self.buckets)
        for b in self.bucket_of:
            self.members[b] += 1
        self.average = average
        self._use_avg = average and dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.reset()

    def reset(self):
        self.pending = list(self.members)
        self.launched = [False] * len(self.buckets)
        self.handles = []

    def _launch(self, b: int):
        if self.launched[b] or self.world == 1:
            self.launched[b] = True
            return
        s, e = self.buckets[b]
        view = self.flat[s:e]
        if self.flat.is_cuda:
            from . import ops
            if ops.WGRAD_STREAM is not None:      # weight gradients of this bucket may still be in flight there
                torch.cuda.current_stream().wait_stream(ops.WGRAD_STREAM)
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        h = dist.all_reduce(view, op=op, group=self.group, async_op=True)
        self.handles.append((h, view))
        self.launched[b] = True

    def mark_ready(self, param_index: int):
        b = self.bucket_of[param_index]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def wait(self):
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for h, view in self.handles:
            h.wait()
            if self.average and not self._use_avg and self.world > 1:
                view.div_(self.world)
        self.reset()
