"""The optimizer step of /root/reference/train.py:113-159 on the HIP path.

    seg_loss(...)     loss assembly of train.py:118-142 (fused single-pass kernels)
    FusedRMSprop      clip_grad_norm_(1.0) + RMSprop(lr, weight_decay=1e-8, momentum=0.999) of
                      train.py:80-81,157-158 as two passes over one flat fp32 buffer
    train_step(...)   zero_grad -> forward (autocast) -> loss -> NaN check -> backward -> clip -> step
    TrainStepper      model + optimizer (+ data-parallel sync) bundle used by bench.py
    train_model(...)  the epoch loop of train.py:29-220 on a user-supplied iterable of batches
A reference-style loop with torch.optim.RMSprop / clip_grad_norm_ on `model.parameters()` also
works unchanged: the modules are ordinary nn.Modules.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, Iterable, Optional

import torch
import torch.nn as nn

from . import dp as dpmod
from . import ops
from ._lib import LIB


# ----------------------------------------------------------------------------------- loss
def seg_loss(masks_pred: torch.Tensor, true_masks: torch.Tensor, n_classes: int, *, reduce_sums=None, world: float = 1,
             boundary_weight: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """train.py:118-142.  masks_pred: logits [B,n_classes,H,W] (as returned by the model);
    true_masks: int64 [B,H,W] with the dataset's values {0,1,2,..} (NOT yet // 2)."""
    if n_classes == 1:
        w_b = 0.25 if boundary_weight is None else boundary_weight          # train.py:134
        lg = masks_pred.squeeze(1)
        total, bce, dice, bnd, nan_flag = ops.SegLossBinaryFn.apply(lg, true_masks, 2, w_b, 51, 15.0, reduce_sums, world)
        terms = {"loss": total, "bce": bce, "dice": dice, "boundary": bnd}
        if nan_flag is not None:
            terms["nan_flag"] = nan_flag          # float 0 / 1 written by the loss's finishing block (train.py:149)
        return terms
    w_b = 0.0 if boundary_weight is None else boundary_weight               # train.py:143-147 is commented out
    lg = masks_pred.permute(0, 2, 3, 1)
    out = ops.SegLossMulticlassFn.apply(lg, true_masks, w_b, 51, 7.0, reduce_sums, world)
    return {"loss": out[0], "ce": out[1].detach(), "dice": out[2].detach(), "boundary": out[3].detach()}


# ----------------------------------------------------------------------------------- optimizer
class FusedRMSprop:
    """clip_grad_norm_ + torch.optim.RMSprop(momentum > 0, centered=False) over ONE flat buffer.

    Every parameter is re-pointed at a view of `flat_p` (values preserved, strides preserved, so
    channels_last weights stay channels_last); gradients are gathered into `flat_g` by
    post-accumulate hooks in backward-ready order, which is also the bucket order of the
    data-parallel all-reduce."""

    def __init__(self, params: Iterable[nn.Parameter], lr: float = 1e-5, alpha: float = 0.99, eps: float = 1e-8,
                 weight_decay: float = 1e-8, momentum: float = 0.999, gradient_clipping: float = 1.0,
                 process_group=None, bucket_bytes: int = 8 << 20):
        plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("FusedRMSprop got no parameters")
        dev = plist[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedRMSprop needs parameters on the GPU (no CPU fallback)")
        self.params = list(reversed(plist))          # backward produces gradients roughly in this order
        self.defaults = dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=momentum)
        self.param_groups = [dict(self.defaults, params=self.params)]
        self.gradient_clipping = float(gradient_clipping) if gradient_clipping else 0.0
        slices, off = [], 0
        for p in self.params:
            n = p.numel()
            slices.append((off, n))
            off += (n + 3) // 4 * 4                  # keep every slice 16-byte aligned
        self.total = off
        self.slices = slices
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_buf = torch.zeros(off, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ws = torch.empty(LIB.query("uh_optim_ws_bytes", off), dtype=torch.uint8, device=dev)
        self._index = {}
        self._grad_keys = []
        self._hooks = []
        self._closed = False
        # one backward per step: which slices of flat_g hold THIS step's gradient (reset by zero_grad / step)
        self._fresh = [False] * len(self.params)
        self._stale_verified = None                  # data parallel: the fresh / stale pattern of the last step (agreed on the first)
        self._probe, self._probe_code = None, None
        with torch.no_grad():
            for i, (p, (o, n)) in enumerate(zip(self.params, slices)):
                if p.dtype != torch.float32:
                    raise RuntimeError("FusedRMSprop expects fp32 master parameters")
                if not _is_dense(p):
                    raise RuntimeError("parameters must be dense (contiguous or channels_last)")
                # a parameter belongs to ONE FusedRMSprop: a stale owner's hook would steal the gradient (it fires first,
                # copies p.grad into its dead buffer and clears it), so an earlier owner is closed here (take-over)
                prev = getattr(p, "_uh_owner", None)
                prev = prev() if prev is not None else None
                if prev is not None and prev is not self:
                    prev.close()
                view = torch.as_strided(self.flat_p, p.shape, p.stride(), o)
                view.copy_(p)
                p.data = view
                self._index[id(p)] = i
                p._uh_owner = weakref.ref(self)
                # backward kernels write this parameter's gradient straight into the flat buffer (ops.GRAD_DST)
                ops.GRAD_DST[view.data_ptr()] = (torch.as_strided(self.flat_g, p.shape, p.stride(), o),
                                                 _weak_callback(self, i))
                self._grad_keys.append(view.data_ptr())
                # the hook holds the optimizer weakly (a bound method would keep it -- and its four flat buffers --
                # alive for as long as the parameter lives) and its handle is kept so that close() can remove it
                self._hooks.append(p.register_post_accumulate_grad_hook(_weak_grad_hook(self)))
        self.sync = None
        if dpmod.sync_enabled(process_group):
            self.sync = dpmod.BucketedGradSync(self.flat_g, slices, bucket_bytes, process_group)

    def close(self):
        """Detach from the parameters: hooks removed, in-place gradient destinations unregistered.  The parameters keep
        their current values (they stay views of this optimizer's flat buffer until someone re-points them)."""
        if getattr(self, "_closed", True):
            return
        self._closed = True
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for k in self._grad_keys:
            ent = ops.GRAD_DST.get(k)
            if ent is not None and ent[0].untyped_storage().data_ptr() == self.flat_g.untyped_storage().data_ptr():
                ops.GRAD_DST.pop(k, None)
        self._grad_keys = []
        for p in self.params:
            ref = getattr(p, "_uh_owner", None)
            if ref is not None and ref() in (self, None):
                p._uh_owner = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _mark(self, i: int, event=None):
        if self._fresh[i]:
            raise RuntimeError("FusedRMSprop: a second gradient for the same parameter arrived before step(): the backward "
                               "kernels overwrite the flat gradient buffer, so gradient accumulation over several "
                               "backward() calls is not supported (call zero_grad() + step() once per backward)")
        self._fresh[i] = True
        if self.sync is not None:
            self.sync.mark_ready(i, event)

    def _written(self, i: int, event=None):
        if not self._closed:
            self._mark(i, event)

    def _on_grad(self, p: torch.Tensor):
        if self._closed:
            return
        i = self._index[id(p)]
        o, n = self.slices[i]
        g = p.grad
        if g is None:            # written in place by the backward kernel (ops.GRAD_DST); readiness came via _written
            return
        torch.as_strided(self.flat_g, p.shape, p.stride(), o).copy_(g)     # parameters without an in-place writer
        p.grad = None
        self._mark(i)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None
        self._fresh = [False] * len(self.params)

    def abort_step(self):
        """Drop the gradients of the current backward without applying them (NaN loss, train.py:149-151): the side
        stream and any gradient all-reduce already in flight are drained so that the next step starts clean."""
        if ops.WGRAD_STREAM is not None:
            torch.cuda.current_stream().wait_stream(ops.WGRAD_STREAM)
        if self.sync is not None:
            self.sync.wait()
        self._fresh = [False] * len(self.params)

    @staticmethod
    def _pattern_code(pattern) -> float:
        """A small integer (exact in fp32, its square times the world size too) that names a fresh / stale pattern."""
        return float(1 + sum((i + 1) * 7919 for i, f in enumerate(pattern) if f) % 4093)

    def pattern_probe(self) -> Optional[torch.Tensor]:
        """Data parallel: [c, c^2] of the pattern the LAST step ran with (device tensor, rewritten only when the pattern changes).
        train_step appends it to the NaN-flag all-reduce every rank issues every step anyway: sum(c^2) * world == sum(c)^2 iff
        every rank had the same pattern."""
        if self.sync is None or self._stale_verified is None:
            return None
        code = self._pattern_code(self._stale_verified)
        if self._probe is None or self._probe_code != code:
            self._probe = torch.tensor([code, code * code], dtype=torch.float32, device=self.flat_g.device)
            self._probe_code = code
        return self._probe

    def _check_stale_pattern(self):
        """Data parallel: "skip a parameter that got no gradient" is only what torch.optim + DDP do when EVERY rank skips it
        (behind DDP's all-reduce .grad is non-None on every rank).  A parameter that is fresh on one rank and stale on another
        would be updated on one replica only, silently.  The pattern is therefore compared across the ranks on the first
        step (one tiny all-reduce + host read, once).  Afterwards it MAY change -- freezing or unfreezing layers mid-run is
        legal with torch.optim + DDP -- as long as it changes on every rank alike: each step's pattern rides, as a checksum, in
        the NEXT step's NaN-flag all-reduce (pattern_probe; a collective every rank issues every step, so nobody can be left
        waiting), and a mismatch raises on all ranks together, one step late.  U-Net uses every parameter on every step."""
        pattern = tuple(self._fresh)
        if self._stale_verified is None:
            import torch.distributed as dist
            bits = torch.tensor([1.0 if f else 0.0 for f in pattern], dtype=torch.float32, device=self.flat_g.device)
            both = torch.stack([bits, -bits])
            dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.sync.group)     # max(bits), -min(bits)
            if bool((both[0] != -both[1]).any().item()):
                # (every rank takes this branch: the buckets are drained and the step is abandoned before anyone raises)
                self.sync.wait()
                self._fresh = [False] * len(self.params)
                raise RuntimeError("FusedRMSprop (data parallel): a parameter received a gradient on some ranks and none on "
                                   "others; the replicas would diverge (give every rank the same set of trainable / used "
                                   "parameters)")
        self._stale_verified = pattern

    def check_probe(self, s1: float, s2: float, world: int):
        """Host side of pattern_probe: s1 = sum of the ranks' codes, s2 = sum of their squares (read with the NaN flag)."""
        if abs(s2 * world - s1 * s1) > 0.5:
            self.abort_step()
            raise RuntimeError("FusedRMSprop (data parallel): in the previous step the set of parameters that received gradients "
                               "differed between the ranks; the replicas have diverged by one update (give every rank the same "
                               "set of trainable / used parameters)")

    @torch.no_grad()
    def step(self):
        if self._closed:
            raise RuntimeError("FusedRMSprop.step() after close() (another FusedRMSprop took the parameters over)")
        ops.flush_slabs()
        if ops.WGRAD_STREAM is not None:
            torch.cuda.current_stream().wait_stream(ops.WGRAD_STREAM)     # weight gradients written on the side stream
        # A parameter that received no gradient this step is SKIPPED, as torch.optim skips parameters whose .grad is None
        # (an unused or frozen parameter must not drift under weight decay / momentum): its slice of the flat gradient is
        # zeroed -- so that it adds nothing to the clipping norm and a data-parallel peer that did produce a gradient still
        # meets a matching all-reduce -- and the update runs over the contiguous runs of fresh slices only.
        stale = [i for i, f in enumerate(self._fresh) if not f]
        if stale and len(stale) == len(self._fresh):
            raise RuntimeError("FusedRMSprop.step() without a backward pass since the last step / zero_grad")
        runs = [(0, self.total)]
        if stale:
            for i in stale:
                o, n = self.slices[i]
                self.flat_g[o:o + n].zero_()
                if self.sync is not None:
                    self.sync.mark_ready(i)
            runs, start = [], None
            for i, (o, n) in enumerate(self.slices):
                end = o + (n + 3) // 4 * 4
                if self._fresh[i]:
                    if start is None:
                        start = o
                    last = end
                elif start is not None:
                    runs.append((start, last - start))
                    start = None
            if start is not None:
                runs.append((start, last - start))
        if self.sync is not None:
            # (behind the stale slices' mark_ready: every bucket has been issued -- in index order on every rank -- so the
            # comparison's own all-reduce sits at the same place in every rank's sequence of collectives)
            self._check_stale_pattern()
        self._fresh = [False] * len(self.params)
        if self.sync is not None:
            self.sync.wait()
        g = self.param_groups[0]
        st = torch.cuda.current_stream().cuda_stream
        LIB.call("uh_grad_sumsq", self.flat_g.data_ptr(), self.total, self.norm.data_ptr(), self.ws.data_ptr(),
                 self.ws.numel(), st)
        for o, n in runs:
            LIB.call("uh_rmsprop_step", self.flat_p[o:].data_ptr(), self.flat_g[o:].data_ptr(), self.flat_sq[o:].data_ptr(),
                     self.flat_buf[o:].data_ptr(), n, self.norm.data_ptr(), self.gradient_clipping, float(g["lr"]),
                     float(g["alpha"]), float(g["eps"]), float(g["weight_decay"]), float(g["momentum"]), st)
        ops.WEIGHT_EPOCH += 1        # parameters changed behind torch's version counters (see ops.packed_w3x3_cached)
        return self.norm

    def grad_of(self, p: torch.Tensor) -> torch.Tensor:
        """The (clipped, after step()) gradient of `p` as stored in the flat buffer."""
        o, n = self.slices[self._index[id(p)]]
        return torch.as_strided(self.flat_g, p.shape, p.stride(), o)


def _weak_grad_hook(opt: "FusedRMSprop"):
    ref = weakref.ref(opt)

    def hook(p):
        o = ref()
        if o is not None:
            o._on_grad(p)
    return hook


def _weak_callback(opt: "FusedRMSprop", i: int):
    ref = weakref.ref(opt)

    def cb(event=None):
        o = ref()
        if o is not None:
            o._written(i, event)
    return cb


def _is_dense(p: torch.Tensor) -> bool:
    sizes_strides = sorted(((st, sz) for sz, st in zip(p.shape, p.stride()) if sz > 1))
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True


# ----------------------------------------------------------------------------------- one step
_ONES = {}
_PINNED = {}


def _one_like(loss: torch.Tensor) -> torch.Tensor:
    key = (loss.device, loss.dtype, tuple(loss.shape))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    return t


def _pinned_flag(device, n: int = 1) -> torch.Tensor:
    """Pinned floats the NaN flag (and, data parallel, the pattern probe) are copied into (one buffer per device: it is read
    back before the next step writes it)."""
    t = _PINNED.get(device)
    if t is None:
        t = _PINNED[device] = torch.zeros(4, dtype=torch.float32).pin_memory()
    return t[:n]


def train_step(model: nn.Module, optimizer, images: torch.Tensor, true_masks: torch.Tensor, *, amp: bool = True,
               gradient_clipping: float = 1.0, reduce_sums=None, world: int = 1, check_nan: bool = True,
               boundary_weight: Optional[float] = None, cc_loss: bool = False) -> Dict[str, torch.Tensor]:
    """Statement sequence of train.py:113-159.  `optimizer` is a FusedRMSprop (clipping fused into
    its step) or any torch.optim optimizer (then clip_grad_norm_ is applied as in the reference)."""
    assert images.shape[1] == model.n_channels, \
        f"Network has been defined with {model.n_channels} input channels, but loaded images have " \
        f"{images.shape[1]} channels. Please check that the images are loaded correctly."   # train.py:108-111
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        masks_pred = model(images)
        terms = seg_loss(masks_pred, true_masks, model.n_classes, reduce_sums=reduce_sums, world=world,
                         boundary_weight=boundary_weight)
    if cc_loss and model.n_classes == 1:
        # the block train.py:124-132 keeps commented out (BASELINE config 5 turns it on): a Python float, no gradient.
        # sigmoid(x) > 0.5 <=> x > 0, so the 0/1 threshold mask stands in for the probabilities.  The masks stay on the
        # device (uh_cc_loss_device); only the resulting scalar is read back, which waits for the forward like the
        # reference's .cpu() does.
        from .utils.connected_component_loss import connected_component_loss
        cc = connected_component_loss(ops.threshold_mask(masks_pred.detach().squeeze(1)), edge_distance=50, min_area=1000,
                                      penalty_weight=0.1)
        terms["cc"] = torch.tensor(cc, device=masks_pred.device)
        terms["loss"] = terms["loss"] + cc
    loss = terms["loss"]
    nan_flag = terms.pop("nan_flag", None)       # (not one of the reported terms)
    nan_host = nan_event = None
    if check_nan:
        # train.py:149-151 raises before backward().  Reading the flag right here would stall the host until the
        # forward has drained and leave the GPU idle while backward is being enqueued, so the flag travels to pinned
        # memory asynchronously and is read after backward has been ENQUEUED, still before the optimizer step: a NaN
        # loss raises the same error and never reaches the parameters.
        flag = nan_flag
        if flag is None or cc_loss:
            flag = torch.isnan(loss.detach()).reshape(1).float()
        probe = None
        if reduce_sums is not None:
            # data parallel: boundary_loss (and so the loss value) is per rank; every rank must take the SAME decision, or
            # the ranks that continue hang at their next collective while one has raised.  The same all-reduce carries the
            # checksum of the previous step's fresh / stale gradient pattern (FusedRMSprop.pattern_probe).
            probe = optimizer.pattern_probe() if isinstance(optimizer, FusedRMSprop) else None
            if probe is not None:
                flag = torch.cat([flag.reshape(1), probe])
            flag = reduce_sums(flag)
        nan_host = _pinned_flag(loss.device, flag.numel())
        nan_host.copy_(flag, non_blocking=True)
        nan_event = torch.cuda.Event()
        nan_event.record()
    optimizer.zero_grad(set_to_none=True)
    # (the seed gradient is a cached constant: loss.backward() alone launches a fill kernel for it every step)
    loss.backward(gradient=_one_like(loss))
    ops.flush_slabs()            # the backward-weights reductions that waited for one batched launch (ops.SlabBatch)
    if nan_event is not None:
        nan_event.synchronize()
        host = nan_host.tolist()
        if len(host) == 3:
            optimizer.check_probe(host[1], host[2], optimizer.sync.world)     # (`world` may be a ragged shard's gb / lb: the group's size counts)
        if host[0] > 0:
            if isinstance(optimizer, FusedRMSprop):
                optimizer.abort_step()
            raise RuntimeError("Fatal: NaN loss detected!")                                   # train.py:149-151
    if isinstance(optimizer, FusedRMSprop):
        terms["grad_norm"] = optimizer.step()
    else:
        terms["grad_norm"] = torch.nn.utils.clip_grad_norm_(model.parameters(), gradient_clipping)
        optimizer.step()
    terms["logits"] = masks_pred.detach()
    return terms


class TrainStepper:
    """Model + FusedRMSprop (+ RCCL gradient sync when torch.distributed is initialised)."""

    def __init__(self, model: nn.Module, lr: float = 1e-5, weight_decay: float = 1e-8, momentum: float = 0.999,
                 gradient_clipping: float = 1.0, amp: bool = True, process_group=None, check_nan: bool = True,
                 wgrad_stream: Optional[bool] = None, cc_loss: bool = False, sync_bn: bool = False, fp32_mode: str = "exact"):
        self.model = model
        # wgrad_stream: backward-weights kernels (and their slab reductions) on a stream of their own, beside the BatchNorm /
        # pool / upsample backward kernels of the layers that follow.  True / False force it; None (default) decides per step
        # (`_side_for`): ON for bf16 steps of at least 2^20 pixels per process.  Round 2 measured one stream 2-3 % faster and
        # turned it off; with round 4's kernels (bf16 slabs, rotated backward-weights loop) it is the other way round on the
        # GPU-bound bf16 configurations -- batch 8: 872.2 -> 884.9 images/s (+1.5 %), batch 4 +2.2 %, batch 32 +1 %, config 4
        # +1.1 %, transposed-conv variant +0.8 %, three interleaved rounds each -- while exact fp32 loses 3 % and batch 2
        # (host-bound: two more stream switches per layer) 7-20 % (DESIGN.md "Measured (round 4)").  UH_SIDE_STREAM=0 / 1
        # overrides the automatic choice.
        # (kept on the instance and installed for the duration of step(): another live stepper keeps its own choice)
        self._side_auto = wgrad_stream is None
        env = os.environ.get("UH_SIDE_STREAM")
        if self._side_auto and env in ("0", "1"):
            wgrad_stream, self._side_auto = (env == "1"), False
        want = wgrad_stream is None or bool(wgrad_stream)
        # (a high-priority stream: 897.6 vs 894.9 images/s on the default priority, 893.2 on a low one, 884.8 on one stream --
        # scratch/r4_prio_probe.py, three interleaved rounds)
        self.wgrad_stream = torch.cuda.Stream(priority=-1) if (want and torch.cuda.is_available()) else None
        self.amp = amp
        self.check_nan = check_nan
        if fp32_mode not in ("exact", "bf16x3"):
            raise ValueError("fp32_mode must be 'exact' or 'bf16x3'")
        self.fp32_mode = fp32_mode          # how fp32 activations are convolved (ops.FP32_MODE); irrelevant under amp
        self.cc_loss = cc_loss
        self.group = process_group
        self.world = dpmod.world_size(process_group)
        self.reduce_sums = dpmod.make_sum_reducer(process_group)
        # sync_bn: BatchNorm statistics (forward) and their backward sums over the GLOBAL batch -> the data-parallel step
        # reproduces the single-process step on the concatenated batch (SURVEY.md 8e option 2); default = per-rank
        # statistics like stock DDP.  One all_gather (2C+1 floats) + one all_reduce (2C floats) per BatchNorm layer.
        # Its collectives run on the GRADIENT group by default.  UH_SYNCBN_OWN_GROUP=1 gives them a process group of their own
        # (on the gradient group they queue, inside RCCL's one stream per communicator, behind bucket all-reduces; with the
        # opt-in side stream for backward-weights those wait for side-stream events, and the critical-path BatchNorm backward
        # would stall behind work it does not depend on).  Opt-in because two communicators issuing collectives concurrently
        # from one process is the pattern RCCL deadlocks on when the ranks enqueue them in different orders, and no N > 1
        # RCCL box has run it yet; because dist.new_group is a collective over the WHOLE default group (a stepper built on a
        # sub-group must be constructed by every rank of the job); and because the extra communicator has to be destroyed
        # (close()).  UH_DP_FORCE_SYNC=1 turns SyncBN on in a ONE-rank group too, so that both communicators are exercised
        # against the real backend on a one-GPU box (tests/test_gpu_dp.py).
        self.sync_bn = None
        self.bn_group = None
        if sync_bn and dpmod.sync_enabled(process_group):
            import torch.distributed as dist
            group = process_group
            if os.environ.get("UH_SYNCBN_OWN_GROUP") == "1":
                ranks = dist.get_process_group_ranks(process_group) if process_group is not None else list(range(dist.get_world_size()))
                self.bn_group = group = dist.new_group(ranks=ranks)
            self.sync_bn = (group, self.world)
        self.optimizer = FusedRMSprop(model.parameters(), lr=lr, weight_decay=weight_decay, momentum=momentum,
                                      gradient_clipping=gradient_clipping, process_group=process_group)
        self._pack = None
        # the closing reductions of backward-weights wait for one batched launch behind the backward pass; data parallel: for
        # one launch per gradient bucket, so that the bucket's all-reduce still starts under the rest of the backward pass
        sync = self.optimizer.sync
        self._slabs = ops.SlabBatch(flush_bytes=None if sync is None else 8 << 20)

    def close(self):
        """Release what the stepper owns outside torch's garbage collection: the optimizer's hooks and, when SyncBN runs on
        a process group of its own (UH_SYNCBN_OWN_GROUP=1), that communicator."""
        opt = getattr(self, "optimizer", None)
        if opt is not None:
            opt.close()
        grp, self.bn_group = getattr(self, "bn_group", None), None
        if grp is not None:
            import torch.distributed as dist
            self.sync_bn = None
            if dist.is_initialized():
                dist.destroy_process_group(grp)

    def __del__(self):
        try:
            if getattr(self, "bn_group", None) is not None:
                self.close()
        except Exception:
            pass

    SIDE_MIN_PIXELS = 1 << 20

    def _side_for(self, images):
        """The stream backward-weights runs on in the step over `images` (None: the launch stream)."""
        if self.wgrad_stream is None or not self._side_auto:
            return self.wgrad_stream
        if self.sync_bn is not None and self.bn_group is None:
            # SyncBN's collectives share the gradient communicator (one RCCL stream): a BatchNorm-backward all_reduce on the critical
            # path would queue behind bucket all-reduces that wait for side-stream events.  Unmeasured over RCCL with more than
            # one rank (one-GPU boxes), so the automatic choice stays on the safe side: one stream.  UH_SYNCBN_OWN_GROUP=1 (a
            # communicator of its own) or an explicit wgrad_stream=True keep the side stream.
            return None
        big = images.is_cuda and images.shape[0] * images.shape[-2] * images.shape[-1] >= self.SIDE_MIN_PIXELS
        return self.wgrad_stream if (self.amp and big) else None

    def step(self, images, true_masks, global_batch: Optional[int] = None):
        """One optimizer step (train.py:113-159).  `global_batch` (data parallel with sync_bn): the sum of the ranks' batch
        sizes when the caller knows it (equal shards: world * B) -- otherwise it is all-reduced here, which costs a blocking
        host read per step."""
        self.model.train()
        side = self._side_for(images)
        ops.WGRAD_STREAM = side
        ops.SYNC_BN = self.sync_bn
        ops.SYNC_BN_BATCH = None
        world = self.world
        if self.sync_bn is not None:
            lb = int(images.shape[0])
            gb = int(global_batch) if global_batch is not None else dpmod.global_batch(lb, self.group, images.device)
            ops.SYNC_BN_BATCH = (gb, lb)
            world = gb / lb               # ragged shards: the loss is normalised by the GLOBAL pixel count n * gb / lb
        fp32_mode_before, ops.FP32_MODE = ops.FP32_MODE, self.fp32_mode        # (a per-step setting: restored below, so that one
        # stepper's bf16x3 products do not leak into whatever runs next in the process)
        # one launch packs every 3x3 filter (bf16/fp32 KRSC + backward-data layout) for this step's forward/backward
        dt = torch.bfloat16 if self.amp else getattr(self.model, "compute_dtype", torch.float32)
        if self._pack is None or self._pack.dtype != dt:
            ws = [m.weight for m in self.model.modules() if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3)]
            self._pack = ops.ConvWeightPack(ws, dt) if ws and all(w.is_cuda for w in ws) else None
        if self._pack is not None:
            ops.WEIGHT_PACK = self._pack
            self._pack.refresh()
        self._slabs.reset()
        ops.SLAB_BATCH = self._slabs if side is None else None
        try:
            return train_step(self.model, self.optimizer, images, true_masks, amp=self.amp,
                              reduce_sums=self.reduce_sums, world=world, check_nan=self.check_nan, cc_loss=self.cc_loss)
        finally:
            ops.SLAB_BATCH = None
            ops.FP32_MODE = fp32_mode_before


class GraphedTrainStepper(TrainStepper):
    """TrainStepper whose whole step (forward, loss, backward, clip + RMSprop) is captured once into a HIP graph and
    replayed per batch: for launch-bound models (UNet_S / UNet_T: ~500 small kernels per step) the host no longer paces
    the GPU.  Fixed batch shape; re-captured when the learning rate changes (it is a kernel argument); single process
    only.  A NaN loss is detected after the replay (the fused optimizer kernel has skipped the update: non-finite norm)."""

    def __init__(self, model: nn.Module, *args, warmup: int = 2, **kw):
        kw.setdefault("wgrad_stream", False)         # one stream unless asked for: the graph is for launch-bound small models
        super().__init__(model, *args, **kw)
        if self.world != 1:
            raise RuntimeError("GraphedTrainStepper is single-process; use TrainStepper with torch.distributed")
        if self.cc_loss:
            raise RuntimeError("connected_component_loss returns a Python float (a device read-back every step): not capturable")
        self._warmup = max(1, warmup)
        self._graph = None
        self._key = None

    def _eager_step(self, images, masks):
        self.model.train()
        side = self._side_for(images)
        ops.WGRAD_STREAM = side
        ops.SYNC_BN = None
        fp32_mode_before, ops.FP32_MODE = ops.FP32_MODE, self.fp32_mode
        dt = torch.bfloat16 if self.amp else getattr(self.model, "compute_dtype", torch.float32)
        if self._pack is None or self._pack.dtype != dt:
            ws = [m.weight for m in self.model.modules() if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3)]
            self._pack = ops.ConvWeightPack(ws, dt) if ws and all(w.is_cuda for w in ws) else None
        if self._pack is not None:
            ops.WEIGHT_PACK = self._pack
            self._pack.refresh()
        self._slabs.reset()
        ops.SLAB_BATCH = self._slabs if side is None else None
        try:
            return train_step(self.model, self.optimizer, images, masks, amp=self.amp, reduce_sums=self.reduce_sums,
                              world=self.world, check_nan=False)
        finally:
            ops.SLAB_BATCH = None
            ops.FP32_MODE = fp32_mode_before

    def _capture(self, images, masks):
        opt = self.optimizer
        # the warm-up steps the capture protocol needs must not count as training: snapshot, warm up, restore
        snap_model = {k: v.clone() for k, v in self.model.state_dict().items()}
        snap_opt = (opt.flat_p.clone(), opt.flat_sq.clone(), opt.flat_buf.clone())
        self._im = images.detach().clone(memory_format=torch.preserve_format)
        self._mk = masks.detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self._warmup):
                self._eager_step(self._im, self._mk)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            opt.flat_p.copy_(snap_opt[0]); opt.flat_sq.copy_(snap_opt[1]); opt.flat_buf.copy_(snap_opt[2])
            for k, v in self.model.state_dict().items():
                if v.data_ptr() < opt.flat_p.data_ptr() or v.data_ptr() >= opt.flat_p.data_ptr() + opt.flat_p.numel() * 4:
                    v.copy_(snap_model[k])                    # buffers (running statistics, num_batches_tracked)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            terms = self._eager_step(self._im, self._mk)
            self._nan = torch.isnan(terms["loss"].detach()).reshape(1)
        self._terms = terms
        self._key = (tuple(images.shape), tuple(masks.shape), float(opt.param_groups[0]["lr"]))

    def step(self, images, true_masks):
        key = (tuple(images.shape), tuple(true_masks.shape), float(self.optimizer.param_groups[0]["lr"]))
        if self._graph is None or key != self._key:
            self._capture(images, true_masks)
        self._im.copy_(images)
        self._mk.copy_(true_masks)
        self._graph.replay()
        ops.WEIGHT_EPOCH += 1
        if self.check_nan and bool(self._nan.item()):
            raise RuntimeError("Fatal: NaN loss detected!")                                   # train.py:149-151
        return self._terms


# ----------------------------------------------------------------------------------- epoch loop
def cosine_warm_restarts_lr(base_lr: float, epoch_arg: float, T_0: int = 4, T_mult: int = 2, eta_min: float = 1e-7):
    """CosineAnnealingWarmRestarts.step(epoch) closed form; train.py:187 passes the Dice score as
    `epoch` -- reproduced, not fixed."""
    e = float(epoch_arg)
    if e >= T_0:
        n = int(math.log(e / T_0 * (T_mult - 1) + 1, T_mult)) if T_mult > 1 else int(e // T_0)
        T_cur = e - T_0 * (T_mult ** n - 1) / (T_mult - 1) if T_mult > 1 else e % T_0
        T_i = T_0 * T_mult ** n
    else:
        T_cur, T_i = e, T_0
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * T_cur / T_i)) / 2


def train_model(model, device, train_batches, val_batches=None, epochs: int = 5, learning_rate: float = 1e-5,
                amp: bool = True, weight_decay: float = 1e-8, momentum: float = 0.999,
                gradient_clipping: float = 1.0, log=None):
    """Epoch loop of train.py:29-220 over in-memory iterables of {'image','mask'} batches (the
    reference's directory dataset / checkpoint cadence / tqdm are outside the hot-path scope)."""
    from .evaluate import evaluate
    stepper = TrainStepper(model, lr=learning_rate, weight_decay=weight_decay, momentum=momentum,
                           gradient_clipping=gradient_clipping, amp=amp)
    history = []
    for epoch in range(1, epochs + 1):
        epoch_loss = 0.0
        for batch in train_batches:
            images = batch["image"].to(device=device, dtype=torch.float32, memory_format=torch.channels_last)
            true_masks = batch["mask"].to(device=device, dtype=torch.long)
            terms = stepper.step(images, true_masks)
            epoch_loss += terms["loss"].item()                                                # train.py:163
        rec = {"epoch": epoch, "loss": epoch_loss}
        if val_batches is not None:
            val_score, _, min_score = evaluate(model, val_batches, device, amp)          # train.py:186 (postprocess=True)
            lr = cosine_warm_restarts_lr(learning_rate, float(val_score))                     # train.py:187
            stepper.optimizer.param_groups[0]["lr"] = lr
            rec.update(val_dice=float(val_score), min_dice=float(min_score), lr=lr)
        history.append(rec)
        if log:
            log(rec)
    return history
