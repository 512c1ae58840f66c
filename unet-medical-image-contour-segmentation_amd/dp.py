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

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def sync_enabled(group=None) -> bool:
    """Do the data-parallel collectives run?  With more than one rank, always.  UH_DP_FORCE_SYNC=1 also runs them in a ONE-rank
    process group: every collective is then the identity, but it goes through the real backend -- the way to execute the
    bucket / stream / handle logic against RCCL on a one-GPU box (tests/test_gpu_dp.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("UH_DP_FORCE_SYNC") == "1"


def make_sum_reducer(group=None):
    """Callable that all-reduces (sum) a small tensor in place; None when not distributed."""
    if not sync_enabled(group):
        return None

    def _reduce(t: torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    return _reduce


def global_batch(local_batch: int, group=None, device=None) -> int:
    """Sum of the ranks' batch sizes (one tiny all-reduce + host read; used per step by SyncBN only)."""
    if world_size(group) == 1:
        return int(local_batch)
    t = torch.tensor([int(local_batch)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


class BucketedGradSync:
    """All-reduce (sum; `average=True` for a mean) a flat gradient buffer in contiguous buckets as their slices become
    ready."""

    def __init__(self, flat_grad: torch.Tensor, slices: Sequence[Tuple[int, int]], bucket_bytes: int = 8 << 20,
                 group=None, average: bool = False):
        self.flat = flat_grad
        self.group = group
        self.world = world_size(group)
        self.active = sync_enabled(group)
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
        self.comm_stream = None
        self.time_exposed = False
        self.exposed = []
        self.reset()

    def reset(self):
        self.pending = list(self.members)
        self.next_launch = 0                             # buckets are launched strictly in index order (see mark_ready)
        self.launched = [False] * len(self.buckets)
        self.side_event = [None] * len(self.buckets)     # latest side-stream event among a bucket's gradients
        self.handles = []

    def _launch(self, b: int):
        if self.launched[b] or not self.active:
            self.launched[b] = True
            return
        s, e = self.buckets[b]
        view = self.flat[s:e]
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        if self.flat.is_cuda:
            # The collective is issued from a stream of its own that waits for (a) everything enqueued on the launch stream
            # up to now (the bucket's BatchNorm / bias / ConvTranspose gradients) and (b) the LAST backward-weights kernel of
            # this bucket on the side stream (an event recorded right behind it): the launch stream itself never waits, so
            # the backward keeps running under the transfer and under the side stream's remaining kernels.
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream()
            ev = torch.cuda.Event()
            ev.record()
            self.comm_stream.wait_event(ev)
            if self.side_event[b] is not None:
                self.comm_stream.wait_event(self.side_event[b])
            with torch.cuda.stream(self.comm_stream):
                h = dist.all_reduce(view, op=op, group=self.group, async_op=True)
        else:
            h = dist.all_reduce(view, op=op, group=self.group, async_op=True)
        self.handles.append((h, view))
        self.launched[b] = True

    def mark_ready(self, param_index: int, event=None):
        """The gradient of parameter `param_index` has been enqueued; `event` (optional) = recorded behind its producer
        when that ran on another stream than the current one."""
        b = self.bucket_of[param_index]
        if event is not None:
            self.side_event[b] = event
        self.pending[b] -= 1
        # Strictly in index order: every rank issues the SAME sequence of collectives whatever order its gradients arrive in --
        # a parameter that gets no gradient on one rank only (its slice is marked ready at the optimizer step) delays the
        # buckets behind it on that rank instead of re-ordering them against its peers' (mis-ordered collectives hang RCCL).
        # Gradients arrive in flat order (backward-ready order IS the bucket order), so this costs no overlap.
        while self.next_launch < len(self.buckets) and self.pending[self.next_launch] == 0:
            self._launch(self.next_launch)
            self.next_launch += 1

    def wait(self):
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        self.next_launch = len(self.buckets)
        timed = self.time_exposed and self.flat.is_cuda and self.active
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for h, view in self.handles:
            h.wait()
            if self.average and not self._use_avg and self.world > 1:
                view.div_(self.world)
        if self.flat.is_cuda and self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.exposed.append((e0, e1))      # no kernel sits between the two: their distance is the stall on the collectives
        self.reset()

    def exposed_ms(self):
        """Per-step time the launch stream spent waiting for the gradient all-reduce (needs time_exposed = True)."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self.exposed]
        self.exposed = []
        return out
