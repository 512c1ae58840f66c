"""Torch-side plumbing over the C ABI (include/unet_hip.h): tensor descriptors -> raw pointers,
and the torch.autograd.Function nodes that put the HIP kernels behind the reference's nn.Module
surface.  Activations travel between nodes as pixel-dense NHWC tensors [B,H,W,C] whose pixel
stride may exceed C (channel slices of a wider buffer); the user-facing modules expose them as
logical NCHW views.  Everything here requires GPU tensors: there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from ._lib import LIB, UH_BF16, UH_F32, UH_F32X3, UH_WFRAG, UH_WFRAG_D

BN_EPS_DEFAULT = 1e-5

# optional per-launch timing (bench.py): (kernel family, algorithmic FLOPs, start event, end event, tag) with
# tag = (direction, B, H, W, input channels, output channels) on the 3x3 conv launches, None elsewhere
PROFILE_ON = False
PROFILE = []


def _variant(Cin_split, Cout, esize):
    ck = 64 // esize
    ok = all(c % ck == 0 for c in Cin_split if c) and Cout % 64 == 0
    if ok:
        return "mfma"
    return "stem" if sum(Cin_split) <= 4 else "generic"


class _Timed:
    def __init__(self, name, flops, tag=None):
        self.name, self.flops, self.tag = name, flops, tag

    def __enter__(self):
        if PROFILE_ON:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if PROFILE_ON:
            self.e1.record()
            PROFILE.append((self.name, self.flops, self.e0, self.e1, self.tag))


# ----------------------------------------------------------------------------- helpers
def _require_gpu(t: torch.Tensor, what: str = "tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{what} is on {t.device}: the HIP path needs GPU tensors (no CPU fallback exists)")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return UH_BF16
    if t.dtype == torch.float32:
        return UH_F32
    raise RuntimeError(f"unsupported activation dtype {t.dtype} (float32 / bfloat16 only)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def compute_dtype(default: torch.dtype = torch.float32) -> torch.dtype:
    """bf16 inside torch.autocast (train.py:116 wraps the forward in autocast), else `default`."""
    if torch.is_autocast_enabled():
        return torch.bfloat16
    return default


def pixel_ld(t: torch.Tensor) -> int:
    """Pixel stride (elements) of an NHWC tensor [B,H,W,C]."""
    B, H, W, C = t.shape
    if W > 1:
        return t.stride(2)
    if H > 1:
        return t.stride(1)
    if B > 1:
        return t.stride(0)
    return max(C, 1)


def is_pixel_dense(t: torch.Tensor) -> bool:
    B, H, W, C = t.shape
    if C > 1 and t.stride(3) != 1:
        return False
    ld = pixel_ld(t)
    if ld < C:
        return False
    if W > 1 and t.stride(2) != ld:
        return False
    if H > 1 and t.stride(1) != W * ld:
        return False
    if B > 1 and t.stride(0) != H * W * ld:
        return False
    return True


def dense_nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return `t` ([B,H,W,C]) if it can be walked with one pixel stride, else a packed copy."""
    return t if is_pixel_dense(t) else t.contiguous()


def to_nhwc(x: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Logical NCHW tensor (any memory format) -> pixel-dense NHWC view/copy in `dtype`."""
    _require_gpu(x, "input")
    v = x.permute(0, 2, 3, 1)
    if dtype is not None and v.dtype != dtype:
        v = v.to(dtype)
    return dense_nhwc(v)


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _empty_like_param(p: torch.Tensor) -> torch.Tensor:
    return torch.empty_strided(p.shape, p.stride(), dtype=torch.float32, device=p.device)


# Optional second HIP stream for backward-weights (set by TrainStepper): wgrad only feeds the optimizer, so it can run
# beside the backward-data conv and the HBM-bound BatchNorm-backward kernels of the layers that follow.
WGRAD_STREAM = None


# FusedRMSprop registers, per parameter storage, the view of its flat gradient buffer that the backward kernels
# should write into directly (no gather copy afterwards).  Keyed by data_ptr of the parameter.
GRAD_DST = {}


def _grad_buffer(p: torch.Tensor, wanted: bool = True):
    """-> (buffer to write the gradient of `p` into, completion callback or None).  With a callback the gradient is
    final in the optimizer's flat buffer: the autograd node returns None for it (no AccumulateGrad clone).
    `wanted=False` (autograd does not need this gradient: a frozen parameter): scratch memory, no callback -- the
    optimizer must see the parameter as one that received no gradient (torch.optim skips those)."""
    ent = GRAD_DST.get(p.data_ptr()) if wanted else None
    if ent is not None and ent[0].shape == p.shape and ent[0].stride() == p.stride():
        return ent
    return _empty_like_param(p), None


def _is_krsc_dense(w: torch.Tensor) -> bool:
    O, I, kh, kw = w.shape
    return w.stride() == (kh * kw * I, 1, kw * I, I)


# ----------------------------------------------------------------------------- raw op wrappers
# How fp32 activations are convolved: "exact" = fp32 MFMA (157 TFLOP/s peak, the parity path); "bf16x3" = products on the
# bf16 matrix pipe with hi/lo splits of both operands (forward / backward-data 3x3 convs of MFMA-aligned layers; ~1e-5
# relative, backward-weights stays exact).  Set by TrainStepper(fp32_mode=...) / bench.py --bf16x3.
FP32_MODE = "exact"


def conv_dt(x: torch.Tensor, C0: int, C1: int, Cout: int, need_dx: bool) -> int:
    """dtype code for the 3x3 conv calls of one layer (pack + forward + backward-data use the SAME code)."""
    if x.dtype == torch.float32 and FP32_MODE == "bf16x3":
        Cin = C0 + C1
        if Cin > 4 and C0 % 16 == 0 and C1 % 16 == 0 and Cout % 64 == 0 and (not need_dx or Cin % 64 == 0):
            return UH_F32X3
    return _dt(x)


# Use fragment-major filter packs (UH_WFRAG) wherever the LDS-DMA MFMA kernel runs (False: KRSC everywhere, for A/B runs).
WFRAG = True


def wfrag_ok(B: int, H: int, W: int, C0: int, C1: int, Cout: int, ld0: int, ld1: int, ldy: int, dt: int) -> bool:
    """May the filter of this conv call be packed fragment-major?  (dt: UH_F32 / UH_BF16; bf16x3 packs are KRSC.)"""
    if not WFRAG or dt not in (UH_F32, UH_BF16):
        return False
    return bool(LIB.query("uh_conv3x3_wfrag_ok", B, H, W, C0, C1, Cout, ld0, ld1, ldy, dt))


def pack_w3x3(weight: torch.Tensor, dtype: torch.dtype, need_dgrad: bool, dt_code: Optional[int] = None,
              frag_f: bool = False, frag_d: bool = False):
    """-> (forward pack, backward-data pack or None); frag_f / frag_d: fragment-major instead of KRSC (the conv call that
    consumes the pack must then pass wfrag=True)."""
    O, I = weight.shape[0], weight.shape[1]
    w32 = weight if weight.dtype == torch.float32 else weight.float()
    wf = torch.empty(O * 9 * I, dtype=dtype, device=weight.device)
    wd = torch.empty(O * 9 * I, dtype=dtype, device=weight.device) if need_dgrad else None
    sO, sI, sH, sW = w32.stride()
    if dt_code is None:
        dt_code = UH_BF16 if dtype == torch.bfloat16 else UH_F32
    flags = (UH_WFRAG if frag_f else 0) | (UH_WFRAG_D if (frag_d and need_dgrad) else 0)
    LIB.call("uh_pack_w3x3", w32.data_ptr(), sO, sI, sH, sW, O, I, wf.data_ptr(), _p(wd), dt_code | flags, _stream())
    return wf, wd


# Bumped by every optimizer step that writes parameters through raw pointers (FusedRMSprop): such writes do not touch
# torch's version counters, and the inference path caches packed filters per parameter.
WEIGHT_EPOCH = 0


def packed_w3x3_cached(weight: torch.Tensor, dtype: torch.dtype, frag: bool = False) -> torch.Tensor:
    """Filter pack (KRSC, or fragment-major with frag=True) for the inference forward, cached on the parameter until it
    changes."""
    try:
        version = weight._version
    except RuntimeError:                      # a temporary created under torch.inference_mode (zero-padded small-width filter)
        return pack_w3x3(weight, dtype, False, frag_f=frag)[0]
    key = (weight.data_ptr(), version, WEIGHT_EPOCH, dtype, tuple(weight.stride()), frag)
    hit = getattr(weight, "_uh_packed", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    wf, _ = pack_w3x3(weight, dtype, False, frag_f=frag)
    try:
        weight._uh_packed = (key, wf)
    except AttributeError:
        pass
    return wf


# Bumped whenever a training-mode forward rewrites BatchNorm running statistics through raw pointers (no torch version
# bump): the inference path caches the eval-mode scale / shift per BatchNorm until then.
BN_STATS_EPOCH = 0


def bn_eval_coeffs_cached(gamma, beta, running_mean, running_var, eps: float, C: int, Cp: int):
    """-> (scale, shift) fp32 views of Cp entries each, the first C real: eval-mode BatchNorm (evaluate.py:30, predict.py:17)
    folded to one multiply-add, cached on the gamma parameter until a parameter or a running statistic changes."""
    try:
        key = (gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), gamma._version,
               beta._version, running_mean._version, running_var._version, WEIGHT_EPOCH, BN_STATS_EPOCH, float(eps), C, Cp)
    except RuntimeError:                      # inference tensors carry no version counter
        key = None
    hit = getattr(gamma, "_uh_bn_eval", None) if key is not None else None
    if hit is not None and hit[0] == key:
        return hit[1][:Cp], hit[1][Cp:]
    g32 = gamma if gamma.dtype == torch.float32 else gamma.float()
    b32 = beta if beta.dtype == torch.float32 else beta.float()
    coef = torch.empty(2 * Cp, dtype=torch.float32, device=gamma.device)
    LIB.call("uh_bn_eval_coeffs", g32.data_ptr(), b32.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), float(eps),
             C, coef.data_ptr(), coef[Cp:].data_ptr(), _stream())
    if key is not None:
        try:
            gamma._uh_bn_eval = (key, coef)
        except AttributeError:
            pass
    return coef[:Cp], coef[Cp:]


class ConvWeightPack:
    """Packed (KRSC forward + flipped/transposed backward-data) copies of ALL 3x3 filters of a model, refreshed by one
    kernel launch per train step (TrainStepper) instead of one launch per layer inside the forward."""

    def __init__(self, weights, dtype: torch.dtype):
        self.weights = [w for w in weights if w.dim() == 4 and w.shape[2:] == (3, 3) and w.dtype == torch.float32 and w.is_cuda]
        self.dtype = dtype
        self.epoch = -1
        self._build()

    def _build(self):
        dev = self.weights[0].device
        rows, off, toff = [], 0, 0
        self.offsets = []
        ck = 32 if self.dtype == torch.bfloat16 else 16          # channels per 64-byte K chunk
        self.frags = []
        for w in self.weights:
            O, I = w.shape[0], w.shape[1]
            sO, sI, sH, sW = w.stride()
            # fragment-major copies where the LDS-DMA MFMA kernel will consume them (uh_conv3x3_wfrag_ok's channel rules;
            # the caller of lookup() states what its conv call needs and packs for itself when that differs)
            ff = WFRAG and O % 64 == 0 and I % ck == 0
            fd = WFRAG and I % 64 == 0 and O % ck == 0
            self.frags.append((ff, fd))
            rows.append([w.data_ptr(), sO, sI, sH, sW, O, I, off, toff, (1 if ff else 0) | (2 if fd else 0)])
            self.offsets.append(off)
            off += O * 9 * I
            toff += ((O + 31) // 32) * ((I + 31) // 32) * 9
        self.total = off
        self.ntiles = toff
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.ptrs = [w.data_ptr() for w in self.weights]
        self.strides = [tuple(w.stride()) for w in self.weights]
        self.wf = torch.empty(self.total, dtype=self.dtype, device=dev)
        self.wd = torch.empty(self.total, dtype=self.dtype, device=dev)
        self.versions = [None] * len(self.weights)
        self.index = {p: i for i, p in enumerate(self.ptrs)}

    def refresh(self):
        if any(w.data_ptr() != p or tuple(w.stride()) != st for w, p, st in zip(self.weights, self.ptrs, self.strides)):
            self._build()                       # a parameter was re-allocated (.to(), load with assign=True, ...)
        LIB.call("uh_pack_w3x3_batched", self.table.data_ptr(), len(self.weights), self.ntiles, self.wf.data_ptr(),
                 self.wd.data_ptr(), UH_BF16 if self.dtype == torch.bfloat16 else UH_F32, _stream())
        self.versions = [w._version for w in self.weights]
        self.epoch = WEIGHT_EPOCH

    def lookup(self, weight: torch.Tensor, dtype: torch.dtype, frag_f: bool = False, frag_d: bool = False):
        """-> (w_fwd, w_dgrad) views if the pack holds the CURRENT value of `weight` in `dtype` in the layouts the caller's
        conv calls need (frag_f / frag_d: fragment-major forward / backward-data copy), else None."""
        if FP32_MODE != "exact" and dtype == torch.float32:
            return None                          # bf16x3 layers pack their own [hi | lo] copies
        i = self.index.get(weight.data_ptr())
        if i is None or dtype != self.dtype or self.epoch != WEIGHT_EPOCH or self.versions[i] != weight._version \
                or self.strides[i] != tuple(weight.stride()) or self.frags[i] != (bool(frag_f), bool(frag_d)):
            return None
        n = weight.shape[0] * 9 * weight.shape[1]
        o = self.offsets[i]
        return self.wf[o:o + n], self.wd[o:o + n]


# set by TrainStepper (one pack per model being trained); ConvBnReluFn falls back to per-layer packing on a miss
WEIGHT_PACK: Optional[ConvWeightPack] = None


def conv3x3_fwd(x0: torch.Tensor, x1: Optional[torch.Tensor], w_packed: torch.Tensor, Cout: int,
                want_stats: bool, dt_code: Optional[int] = None, wfrag: bool = False):
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    dt = _dt(x0) if dt_code is None else dt_code
    y = torch.empty((B, H, W, Cout), dtype=x0.dtype, device=x0.device)
    stats, nslab = None, 0
    if want_stats:
        nslab = LIB.query("uh_conv3x3_stat_slabs", B, H, W, C0 + C1, Cout, dt)
        stats = torch.empty(nslab * (2 * Cout + 2), dtype=torch.float32, device=x0.device)
    name = "conv3x3_fwd_" + _variant((C0, C1), Cout, x0.element_size())
    with _Timed(name, 2.0 * B * H * W * Cout * 9 * (C0 + C1), ("fwd" if want_stats else "dgrad", B, H, W, C0 + C1, Cout)):
        LIB.call("uh_conv3x3_fwd", x0.data_ptr(), C0, pixel_ld(x0), _p(x1), C1, 0 if x1 is None else pixel_ld(x1),
                 w_packed.data_ptr(), y.data_ptr(), Cout, Cout, _p(stats), B, H, W, dt | (UH_WFRAG if wfrag else 0), _stream())
    return y, stats, nslab


class SlabBatch:
    """Backward-weights reductions that wait for ONE launch (uh_conv3x3_wgrad_partials + uh_slab_reduce_batched).  A conv layer's
    filter gradient only feeds the optimizer, so its closing reduction over the pixel splits -- one 8-14 us launch per layer, in
    the middle of the backward stream -- is queued here instead, with the callback that tells the optimizer (and the gradient
    all-reduce) that the parameter's gradient is final, and flush() runs them all.  The partial results live in workspaces this
    object owns, one per destination (they are written in one step and read at the flush of the same step).
    `flush_bytes`: flush as soon as that many bytes of gradients are pending (data parallel: one bucket's worth, so that the
    bucket's all-reduce can start under the rest of the backward pass); None: only when asked (TrainStepper: after backward)."""

    def __init__(self, flush_bytes: Optional[int] = None):
        self.flush_bytes = flush_bytes
        self.arena = {}                      # one workspace per destination, kept between steps (up to ~37 MB each: close() frees them)
        self.rows, self.callbacks = [], []
        self.pending_bytes = 0
        self._tables = {}                    # row set -> (device table, rows, blocks); data parallel flushes one set per bucket

    def workspace(self, key, nbytes: int, device) -> torch.Tensor:
        ws = self.arena.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = self.arena[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._tables.clear()             # a table holds workspace addresses
        return ws

    def add(self, desc, callback):
        self.rows.append(tuple(desc))
        self.callbacks.append(callback)
        self.pending_bytes += 4 * int(desc[2])
        if self.flush_bytes is not None and self.pending_bytes >= self.flush_bytes:
            self.flush()

    def reset(self):
        self.rows, self.callbacks, self.pending_bytes = [], [], 0

    def close(self):
        """Drop the workspaces and the device tables (the next use allocates them again)."""
        self.reset()
        self.arena.clear()
        self._tables.clear()

    def flush(self):
        if not self.rows:
            return
        key = tuple(self.rows)
        entry = self._tables.get(key)
        if entry is None:
            # (same layers, same buffers every step: a table is uploaded once per row set -- through pinned memory, without
            # blocking the host in the middle of backward -- and a captured graph replays with it)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("SlabBatch: the reduction table would be built (allocation + host-to-device copy) during graph "
                                   "capture; run one eager step first")
            tab, off = [], 0
            for r in self.rows:
                tab.append([r[0], r[1], r[2], r[3], r[4], r[5], off, r[7]])
                off += r[6]
            dev = next(iter(self.arena.values())).device
            host = torch.tensor(tab, dtype=torch.int64).pin_memory()
            entry = self._tables[key] = (host.to(dev, non_blocking=True), len(tab), off, host)     # (host kept until the copy has run)
        table, nrows, blocks = entry[:3]
        with _Timed("slab_reduce_batched", 0.0):
            LIB.call("uh_slab_reduce_batched", table.data_ptr(), nrows, blocks, _stream())
        cbs = self.callbacks
        self.reset()
        for cb in cbs:
            if cb is not None:
                cb(None)


# set by TrainStepper for the duration of a step; None = every backward-weights call reduces its own slabs at once
SLAB_BATCH: Optional[SlabBatch] = None
# OFF by default -- measured (round 4, one MI355X, config 2, three interleaved rounds, bf16 slabs): 873.1 images/s with the
# per-layer launches, 872.2 with one batched launch behind the backward pass (B=4: 763.0 / 762.2).  The eighteen launches it
# removes are bandwidth-bound (37 MB each at 5.5 TB/s), not latency-bound, and a layer's slabs are read back from the Infinity
# Cache when the reduction runs at once, from HBM when it runs a millisecond later.  UH_DEFER_SLABS=1 turns it on.
DEFER_SLABS = os.environ.get("UH_DEFER_SLABS", "0") == "1"


def flush_slabs():
    if SLAB_BATCH is not None:
        SLAB_BATCH.flush()


def conv3x3_wgrad(dy: torch.Tensor, x0: torch.Tensor, x1: Optional[torch.Tensor], out_krsc: torch.Tensor,
                  split: bool = False, defer_cb=None) -> bool:
    """dW (fp32 KRSC) of a 3x3 conv.  `defer_cb` (a callable taking one argument): the caller allows the closing reduction to be
    queued in ops.SLAB_BATCH; the callback then fires when it has run.  -> True when the result (and the callback) were deferred."""
    B, H, W, Cout = dy.shape
    C0 = x0.shape[3]
    C1 = 0 if x1 is None else x1.shape[3]
    dt = _dt(dy)
    nbytes = LIB.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, C0 + C1, Cout, dt)
    if split and dt == UH_F32 and C0 % 64 == 0 and C1 % 64 == 0 and Cout % 64 == 0:
        dt = UH_F32X3                          # bf16x3 products (ops.FP32_MODE); same workspace as the fp32 plan
    name = "conv3x3_wgrad_" + ("mfma" if (C0 % 64 == 0 and C1 % 64 == 0 and Cout % 64 == 0) else
                                ("stem" if C0 + C1 <= 4 else "generic"))
    batch = SLAB_BATCH if (defer_cb is not None and DEFER_SLABS and name == "conv3x3_wgrad_mfma") else None
    if batch is not None:
        ws = batch.workspace(out_krsc.data_ptr(), nbytes, dy.device)
        desc = (ctypes.c_int64 * 8)()
        with _Timed(name, 2.0 * B * H * W * Cout * 9 * (C0 + C1), ("wgrad", B, H, W, C0 + C1, Cout)):
            LIB.call("uh_conv3x3_wgrad_partials", dy.data_ptr(), pixel_ld(dy), x0.data_ptr(), C0, pixel_ld(x0), _p(x1), C1,
                     0 if x1 is None else pixel_ld(x1), out_krsc.data_ptr(), Cout, ws.data_ptr(), nbytes, B, H, W, dt,
                     ctypes.addressof(desc), _stream())
        if desc[3] == 0:                       # a path without slabs: the gradient is final
            return False
        batch.add(list(desc), defer_cb)
        return True
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    with _Timed(name, 2.0 * B * H * W * Cout * 9 * (C0 + C1), ("wgrad", B, H, W, C0 + C1, Cout)):
        LIB.call("uh_conv3x3_wgrad", dy.data_ptr(), pixel_ld(dy), x0.data_ptr(), C0, pixel_ld(x0), _p(x1), C1,
                 0 if x1 is None else pixel_ld(x1), out_krsc.data_ptr(), Cout, ws.data_ptr(), nbytes, B, H, W, dt, _stream())
    return False


def conv3x3_wgrad_pre(dy: torch.Tensor, x0_raw: torch.Tensor, pre_coef: torch.Tensor, out_krsc: torch.Tensor):
    """Backward-weights of a layer whose input is max(x0_raw * scale + shift, 0) (BatchNorm + ReLU of the producer, applied by
    the loader; pre_coef = [scale | shift | ...] of C0 entries each)."""
    B, H, W, Cout = dy.shape
    C0 = x0_raw.shape[3]
    dt = _dt(dy)
    nbytes = LIB.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, C0, Cout, dt)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    with _Timed("conv3x3_wgrad_mfma", 2.0 * B * H * W * Cout * 9 * C0):
        LIB.call("uh_conv3x3_wgrad_pre", dy.data_ptr(), pixel_ld(dy), x0_raw.data_ptr(), C0, pixel_ld(x0_raw),
                 pre_coef.data_ptr(), pre_coef[C0:].data_ptr(), out_krsc.data_ptr(), Cout, ws.data_ptr(), nbytes, B, H, W, dt,
                 _stream())


def bench_double_conv(B: int, H: int, W: int, Cin: int, Cout: int, dtype: torch.dtype, iters: int = 20):
    """Time the conv kernels of one DoubleConv(Cin -> Cout -> Cout) in isolation: preallocated buffers, the C ABI
    called back to back (`iters` launches between two HIP events on the launch stream, so the figure is kernel
    time, not Python time).  Returns ms and TFLOP/s per kernel; bench.py reports it for the layer the north-star's
    MFMA target is quoted on (the 256-channel DoubleConv)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(0)
    # forward inputs are (BatchNorm -> ReLU) outputs in the network: half zeros, like here (what the operands toggle decides
    # the clock the chip holds under MFMA load); `dyv`, the gradient operand, is dense
    x = torch.relu(torch.randn(B, H, W, Cin, generator=g)).to(dev, dtype)
    h = torch.relu(torch.randn(B, H, W, Cout, generator=g)).to(dev, dtype)
    dyv = torch.randn(B, H, W, Cout, generator=g).to(dev, dtype)
    w1 = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    w2 = (torch.randn(Cout, Cout, 3, 3, generator=g) / (3 * Cout ** 0.5)).to(dev)
    dt = _dt(x)
    fr = {"f1": wfrag_ok(B, H, W, Cin, 0, Cout, Cin, 0, Cout, dt), "d1": wfrag_ok(B, H, W, Cout, 0, Cin, Cout, 0, Cin, dt),
          "f2": wfrag_ok(B, H, W, Cout, 0, Cout, Cout, 0, Cout, dt)}
    w1f, w1d = pack_w3x3(w1, dtype, True, frag_f=fr["f1"], frag_d=fr["d1"])
    w2f, w2d = pack_w3x3(w2, dtype, True, frag_f=fr["f2"], frag_d=fr["f2"])
    yo = torch.empty(B, H, W, Cout, dtype=dtype, device=dev)
    xo = torch.empty(B, H, W, Cin, dtype=dtype, device=dev)
    nslab = LIB.query("uh_conv3x3_stat_slabs", B, H, W, Cin, Cout, dt)
    stats = torch.empty(nslab * (2 * Cout + 2), dtype=torch.float32, device=dev)
    dw = torch.empty(Cout * 9 * Cout, dtype=torch.float32, device=dev)
    wsb = max(LIB.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Cout, Cout, dt),
              LIB.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Cin, Cout, dt))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = _stream()

    def fwd(src, cin, wp, dst, cout, stat, frag=False):
        LIB.call("uh_conv3x3_fwd", src.data_ptr(), cin, cin, None, 0, 0, wp.data_ptr(), dst.data_ptr(), cout, cout,
                 _p(stat), B, H, W, dt | (UH_WFRAG if frag else 0), st)

    def wgrad(dy, src, cin):
        LIB.call("uh_conv3x3_wgrad", dy.data_ptr(), Cout, src.data_ptr(), cin, cin, None, 0, 0, dw.data_ptr(), Cout,
                 ws.data_ptr(), wsb, B, H, W, dt, st)

    def timeit(fn, flops):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        return {"ms": round(ms, 4), "tflops": round(flops / (ms * 1e-3) / 1e12, 1)}

    f1 = 2.0 * B * H * W * Cout * 9 * Cin
    f2 = 2.0 * B * H * W * Cout * 9 * Cout
    out = {
        "shape": f"B{B} {H}x{W} {Cin}->{Cout}->{Cout} {str(dtype)[6:]}",
        "fwd_conv1": timeit(lambda: fwd(x, Cin, w1f, yo, Cout, stats, fr["f1"]), f1),
        "fwd_conv2": timeit(lambda: fwd(h, Cout, w2f, yo, Cout, stats, fr["f2"]), f2),
        "dgrad_conv2": timeit(lambda: fwd(dyv, Cout, w2d, yo, Cout, None, fr["f2"]), f2),
        "dgrad_conv1": timeit(lambda: fwd(dyv, Cout, w1d, xo, Cin, None, fr["d1"]), f1),
        "wgrad_conv2": timeit(lambda: wgrad(dyv, h, Cout), f2),
        "wgrad_conv1": timeit(lambda: wgrad(dyv, x, Cin), f1),
    }
    tot_f = 3 * f2 + 3 * f1
    tot_ms = sum(v["ms"] for k, v in out.items() if isinstance(v, dict))
    out["all_six"] = {"ms": round(tot_ms, 4), "tflops": round(tot_f / (tot_ms * 1e-3) / 1e12, 1)}
    return out


# ----------------------------------------------------------------------------- conv + BN + ReLU
# (process group, world size) when BatchNorm statistics are to be those of the GLOBAL batch of a data-parallel job
# (TrainStepper(sync_bn=True)); None = per-rank statistics (what stock DDP does).
SYNC_BN = None
# (global batch, local batch) of the current step when SYNC_BN is on; None = equal shards
SYNC_BN_BATCH = None


def _sync_bn_forward(coef, m2, n_local, Cout, g32, b32, running_mean, running_var, nbt_ptr, momentum, eps):
    """Merge the per-rank (count, mean, M2) rows into the global-batch statistics: one all_gather of 2C+1 floats per
    layer, then the same uh_bn_finalize over `world` rows (Chan merge in double), which also updates the running
    statistics with the global unbiased variance.  Returns the global pixel count."""
    import torch.distributed as dist
    group, world = SYNC_BN
    dev = coef.device
    mean = coef[2 * Cout:3 * Cout]
    row = torch.cat([mean, m2, torch.full((1,), float(n_local), dtype=torch.float32, device=dev)])
    gathered = torch.empty(world * (2 * Cout + 1), dtype=torch.float32, device=dev)
    dist.all_gather(list(gathered.view(world, 2 * Cout + 1).unbind(0)), row, group=group)
    g2 = gathered.view(world, 2 * Cout + 1)
    stats = torch.empty(world * (2 * Cout + 2), dtype=torch.float32, device=dev)
    stats[:world * 2 * Cout].view(world, 2 * Cout).copy_(g2[:, :2 * Cout])
    stats[world * 2 * Cout:world * 2 * Cout + world].copy_(g2[:, 2 * Cout])
    # the global pixel count: this rank's count scaled by global batch / local batch (TrainStepper all-reduces the batch
    # sizes once per step, so ragged shards -- the last batch of an epoch -- get the right variance denominator)
    gb, lb = SYNC_BN_BATCH if SYNC_BN_BATCH is not None else (world, 1)
    n_total = int(n_local) * gb // lb
    scale, shift, rstd = coef[:Cout], coef[Cout:2 * Cout], coef[3 * Cout:]
    LIB.call("uh_bn_finalize", stats.data_ptr(), world, Cout, n_total, g32.data_ptr(), b32.data_ptr(),
             _p(running_mean), _p(running_var), nbt_ptr, float(momentum), float(eps), scale.data_ptr(), shift.data_ptr(),
             mean.data_ptr(), rstd.data_ptr(), None, _stream())
    return n_total


# Fuse BatchNorm + ReLU of a DoubleConv's last conv with its consumer (csrc/bn_fused.hip) in training: TAIL_POOL where the
# activation is skip connection + max-pool input (every encoder level), TAIL_HEAD where it only feeds the 1x1 OutConv.
# False: the separate kernels (A/B measurements; the tests compare both).
FUSE_TAILS = True
TAIL_NONE, TAIL_POOL, TAIL_HEAD, TAIL_UP = 0, 1, 2, 3


def pool_tail_ok(x0: torch.Tensor, Cout: int) -> bool:
    B, H, W, _ = x0.shape
    return bool(FUSE_TAILS and LIB.query("uh_bn_relu_pool_ok", B, H, W, Cout, _dt(x0)))


FUSE_UP_TAIL = os.environ.get("UH_FUSE_UP_TAIL", "1") != "0"      # (A/B switch of the up-sampling tail alone)


def up_tail_ok(x0: torch.Tensor, Cout: int, Ho: int, Wo: int) -> bool:
    """May (BatchNorm -> ReLU -> nn.Upsample(2, bilinear) -> F.pad to Ho x Wo) run as one kernel behind this layer's conv?"""
    B, H, W, _ = x0.shape
    return bool(FUSE_TAILS and FUSE_UP_TAIL and LIB.query("uh_bn_relu_upsample2x_ok", B, H, W, Cout, Ho, Wo, _dt(x0)))


def head_tail_ok(x0: torch.Tensor, Cout: int, head_weight: torch.Tensor) -> bool:
    return bool(FUSE_TAILS and head_weight.shape[1] == Cout and tuple(head_weight.shape[2:]) == (1, 1) and
                LIB.query("uh_bn_relu_head_ok", Cout, head_weight.shape[0], _dt(x0)))


# BatchNorm + ReLU BETWEEN the two convs of a DoubleConv applied by the second conv's loaders (uh_conv3x3_fwd_pre /
# uh_conv3x3_wgrad_pre): the first layer's node is asked to `defer` its activation (it returns its raw conv output + the
# BatchNorm coefficients), the second takes them as `pre_coef`; the activation is never stored.  bf16 training, 64-aligned
# layers of <= 512 mid channels; bit-identical to the stored-activation path (tests/test_gpu_pre_fusion.py).
# OFF by default -- measured (round 3, one MI355X, config 2, A/B inside one gpurun call): the nine uh_bn_relu_apply launches it
# removes cost 0.37 ms / step and 2.0 GB of traffic; rebuilding the activation in LDS costs the nine forward convs +0.15 ms and
# the nine backward-weights convs +1.0 ms (two workgroups own all of LDS there, so the coefficients travel by ds_bpermute,
# and the rewrite runs once per 8x16 tile whose halo is 1.4x its pixels): 798 -> 735 images/s.  UH_FUSE_PRE=1 turns it on.
FUSE_PRE = os.environ.get("UH_FUSE_PRE", "0") == "1"


def pre_fuse_ok(x0: torch.Tensor, mid: int, Cout: int) -> bool:
    """May the (BatchNorm -> ReLU) between a DoubleConv's convs be applied by the second conv's loader?  x0: the block's
    NHWC input (gives batch, extent, dtype); mid / Cout: channels of the activation in question and of the second conv."""
    if not FUSE_PRE or x0.dtype != torch.bfloat16:
        return False
    B, H, W, _ = x0.shape
    return bool(LIB.query("uh_conv3x3_pre_ok", B, H, W, mid, Cout, mid, Cout, UH_BF16))


# ---- BatchNorm-backward sums formed by the NEXT conv's backward-data.  Inside a DoubleConv the
# gradient of the activation between the two convs is produced by the second conv's backward-data and read back at once by
# uh_bn_relu_bwd_reduce (that tensor + the first conv's raw output).  uh_conv3x3_dgrad_bnsum forms the two per-channel sums in the
# conv epilogue instead -- accumulators still in registers, one read of the raw output -- and the reduce pass is not launched
# (8 of the 12 plain BatchNorm layers of the bilinear UNet).  UH_FUSE_BNSUM=0 turns it off.
FUSE_BNSUM = os.environ.get("UH_FUSE_BNSUM", "1") != "0"
# Layers (image height x channels of the tensor whose BatchNorm sums are formed, e.g. "32x512,512x64") that keep the separate
# reduce pass although the fusion is on -- per-layer A/B runs (scratch/r4_bnsum_layers.sh) and the default exclusion list.
BNSUM_OFF = {tuple(int(v) for v in k.split("x")) for k in os.environ.get("UH_BNSUM_OFF", "").split(",") if k}


class BnSumLink:
    """What ties the two ConvBnReluFn nodes of a DoubleConv together for that fusion: the first layer publishes its raw conv
    output and BatchNorm coefficients in forward; the second layer's backward leaves the partial sums (and the gradient tensor
    they belong to); the first layer's backward uses them if the gradient it receives IS that tensor (autograd hands over the
    same storage when the activation had no other consumer), else it runs the reduce pass as before."""
    __slots__ = ("y", "coef", "dz", "dz_version", "partials", "rows")

    def __init__(self):
        self.y = self.coef = self.dz = self.partials = None
        self.rows = 0
        self.dz_version = -1


class ConvBnReluFn(Function):
    """(nn.Conv2d(3x3, pad 1, no bias) -> nn.BatchNorm2d -> nn.ReLU) of unet_parts.py:15-17 / 18-20 as
    one autograd node.  Inputs: x0 (+ optional x1 = second half of the channel concat of
    unet_parts.py:95, never materialised), the reference-layout parameters, BN buffers.

    `tail` (training only): TAIL_POOL -> returns (z, maxpool2(z)) (unet_parts.py:32 on top), the backward takes
    (dskip, dpool) and never materialises their sum; TAIL_HEAD -> returns the fp32 logits of the 1x1 OutConv
    (head_w [ncls,Cout,1,1], head_b; unet_parts.py:103) and z is never written; TAIL_UP (up_size = (Ho, Wo)) -> returns the
    bilinear x2 up-sampling of z zero-padded to Ho x Wo (unet_parts.py:70,80,85-88: z's only reader) and z is never written."""

    @staticmethod
    def forward(ctx, x0, x1, weight, gamma, beta, running_mean, running_var, num_batches_tracked,
                training: bool, momentum: float, eps: float, tail: int = 0, head_w=None, head_b=None,
                defer: bool = False, pre_coef=None, bnsum_pub: Optional[BnSumLink] = None,
                bnsum_use: Optional[BnSumLink] = None, up_size=None):
        """`bnsum_pub` / `bnsum_use` (BnSumLink): this layer is the first / the second conv of a DoubleConv whose
        BatchNorm-backward sums may be formed by the second conv's backward-data.
        `defer` (training, no tail): the BatchNorm + ReLU of THIS layer is left to its consumer -- returns (y, coef): the raw
        conv output standing in for the activation (its gradient is the activation's gradient) and the [scale | shift | mean
        | rstd] coefficients.  `pre_coef`: x0 is such a raw output; its BatchNorm + ReLU is applied by this layer's conv
        loaders (forward and backward-weights), the activation is never stored."""
        _require_gpu(x0, "activation")
        x0 = dense_nhwc(x0)
        x1 = None if x1 is None else dense_nhwc(x1)
        B, H, W, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[3]
        Cout, Cin = weight.shape[0], weight.shape[1]
        if Cin != C0 + C1:
            raise RuntimeError(f"conv expects {Cin} input channels, got {C0}+{C1}")
        if pre_coef is not None and (x1 is not None or not training or pre_coef.numel() != 4 * C0):
            raise RuntimeError("ConvBnReluFn: a deferred BatchNorm+ReLU input needs a single-source training-mode layer")
        if defer and (not training or tail != TAIL_NONE):
            raise RuntimeError("ConvBnReluFn: defer is a training-mode option of layers without a fused tail")
        need_dx = any(ctx.needs_input_grad[:2])
        cdt = conv_dt(x0, C0, C1, Cout, need_dx and training)
        # fragment-major filter packs where the LDS-DMA MFMA kernel runs (forward: this call; backward-data: the conv of dy
        # [B,H,W,Cout] with the transposed filter into dx [B,H,W,Cin])
        frag_f = wfrag_ok(B, H, W, C0, C1, Cout, pixel_ld(x0), 0 if x1 is None else pixel_ld(x1), Cout, cdt)
        frag_d = bool(need_dx and training) and wfrag_ok(B, H, W, Cout, 0, Cin, Cout, 0, Cin, cdt)
        if training:
            hit = WEIGHT_PACK.lookup(weight, x0.dtype, frag_f, frag_d) if (WEIGHT_PACK is not None and cdt != UH_F32X3) else None
            wf, wd = hit if hit is not None else pack_w3x3(weight, x0.dtype, need_dx, cdt, frag_f, frag_d)
        elif cdt == UH_F32X3:
            wf, wd = pack_w3x3(weight, x0.dtype, False, cdt)[0], None
        else:
            wf, wd = packed_w3x3_cached(weight, x0.dtype, frag_f), None
        ctx.cdt = cdt
        ctx.frag_d = frag_d
        dev = x0.device
        n = B * H * W
        if training:
            if pre_coef is None:
                y, stats, nslab = conv3x3_fwd(x0, x1, wf, Cout, True, cdt, frag_f)
            else:
                y = torch.empty((B, H, W, Cout), dtype=x0.dtype, device=dev)
                nslab = LIB.query("uh_conv3x3_stat_slabs", B, H, W, C0, Cout, cdt)
                stats = torch.empty(nslab * (2 * Cout + 2), dtype=torch.float32, device=dev)
                with _Timed("conv3x3_fwd_mfma", 2.0 * B * H * W * Cout * 9 * C0):
                    LIB.call("uh_conv3x3_fwd_pre", x0.data_ptr(), C0, pixel_ld(x0), pre_coef.data_ptr(), pre_coef[C0:].data_ptr(),
                             wf.data_ptr(), y.data_ptr(), Cout, Cout, stats.data_ptr(), B, H, W,
                             cdt | (UH_WFRAG if frag_f else 0), _stream())
            coef, n_total = _bn_train_coefficients(stats, nslab, Cout, n, gamma, beta, running_mean, running_var,
                                                   num_batches_tracked, momentum, eps)
            scale, shift, mean, rstd = coef[:Cout], coef[Cout:2 * Cout], coef[2 * Cout:3 * Cout], coef[3 * Cout:]
        else:
            # inference (model.eval(): evaluate.py:30, predict.py:17): running statistics -> per-channel scale/shift,
            # applied with the ReLU inside the conv epilogue; nothing is kept for a backward pass
            scale, shift = bn_eval_coeffs_cached(gamma, beta, running_mean, running_var, eps, Cout, Cout)
            z = torch.empty(B, H, W, Cout, dtype=x0.dtype, device=dev)
            LIB.call("uh_conv3x3_fwd_affine_relu", x0.data_ptr(), C0, pixel_ld(x0), _p(x1), C1,
                     pixel_ld(x1) if x1 is not None else 0, wf.data_ptr(), z.data_ptr(), Cout, Cout, scale.data_ptr(),
                     shift.data_ptr(), B, H, W, cdt | (UH_WFRAG if frag_f else 0), _stream())
            ctx.training = False
            if tail != TAIL_NONE:
                raise RuntimeError("ConvBnReluFn: fused tails are a training-mode path")
            return z
        ctx.bn_params = (gamma, beta)
        ctx.training = training
        ctx.dims = (B, H, W, C0, C1, Cout)
        ctx.n_total = n_total
        ctx.sync_bn = SYNC_BN                    # (None = per-rank statistics; a one-rank group under UH_DP_FORCE_SYNC still runs the collectives)
        ctx.tail = tail
        ctx.pre = pre_coef is not None
        ctx.bnsum_pub = ctx.bnsum_use = None
        if FUSE_BNSUM and x0.dtype == torch.bfloat16 and cdt == UH_BF16:
            if bnsum_pub is not None and tail == TAIL_NONE and not defer:
                bnsum_pub.y, bnsum_pub.coef = y, coef
                bnsum_pub.partials = bnsum_pub.dz = None      # (left over when a previous backward never reached this layer)
                ctx.bnsum_pub = bnsum_pub
            if bnsum_use is not None and bnsum_use.y is not None and x1 is None and pre_coef is None:
                ctx.bnsum_use = bnsum_use
        if pre_coef is not None:
            x1 = pre_coef                      # rides in the saved-tensor slot of the (absent) second source
        if defer:
            ctx.save_for_backward(x0, x1, y, coef, wd, weight)
            ctx.mark_non_differentiable(coef)
            return y, coef
        if tail == TAIL_HEAD:
            ncls = head_w.shape[0]
            hw2 = head_w.reshape(ncls, Cout).contiguous().float()
            hb2 = head_b.contiguous().float()
            logits = torch.empty((B, H, W, ncls), dtype=torch.float32, device=dev)
            LIB.call("uh_bn_relu_head_fwd", y.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), hw2.data_ptr(),
                     hb2.data_ptr(), logits.data_ptr(), n, Cout, ncls, _dt(y), _stream())
            ctx.save_for_backward(x0, x1, y, coef, wd, weight, hw2)
            ctx.head_params = (head_w, head_b)
            return logits
        if tail == TAIL_UP:
            Ho, Wo = up_size
            pt, pl = _pad_geometry(H, W, Ho, Wo)
            u = torch.empty((B, Ho, Wo, Cout), dtype=y.dtype, device=dev)
            LIB.call("uh_bn_relu_upsample2x_fwd", y.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), u.data_ptr(), Cout,
                     B, H, W, Cout, Ho, Wo, pt, pl, _dt(y), _stream())
            ctx.up = (Ho, Wo, pt, pl)
            ctx.save_for_backward(x0, x1, y, coef, wd, weight)
            return u
        z = torch.empty_like(y)
        if tail == TAIL_POOL:
            pooled = torch.empty((B, H // 2, W // 2, Cout), dtype=y.dtype, device=dev)
            LIB.call("uh_bn_relu_pool_apply", y.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), z.data_ptr(), Cout,
                     pooled.data_ptr(), Cout, B, H, W, Cout, _dt(y), _stream())
            ctx.save_for_backward(x0, x1, y, coef, wd, weight)
            ctx.set_materialize_grads(False)
            return z, pooled
        LIB.call("uh_bn_relu_apply", y.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), z.data_ptr(), Cout,
                 n, Cout, _dt(y), _stream())
        ctx.save_for_backward(x0, x1, y, coef, wd, weight)
        return z

    @staticmethod
    def backward(ctx, *grads):
        tail = ctx.tail if ctx.training else TAIL_NONE
        hw2 = None
        if tail == TAIL_HEAD:
            x0, x1, y, coef, wd, weight, hw2 = ctx.saved_tensors
        else:
            x0, x1, y, coef, wd, weight = ctx.saved_tensors
        B, H, W, C0, C1, Cout = ctx.dims
        if not ctx.training:
            raise RuntimeError("backward through eval-mode BatchNorm is not part of the train path")
        pre_coef = None
        if ctx.pre:
            pre_coef, x1 = x1, None
        Cin = C0 + C1
        n = B * H * W
        dev = y.device
        scale, shift, mean, rstd = coef[:Cout], coef[Cout:2 * Cout], coef[2 * Cout:3 * Cout], coef[3 * Cout:]
        dt = _dt(y)
        nblk = LIB.query("uh_bn_bwd_nblk", n, Cout)
        partials = torch.empty(nblk * 2 * Cout, dtype=torch.float32, device=dev)
        bn_args = (y.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr())
        dhead = (None, None)
        # The gradient of z: a tensor (no tail), dlogits . head_w (head tail) or dskip + route(dpool) (pool tail) -- the
        # last two are rebuilt inside both BatchNorm passes instead of being stored.  `reduce()` writes the per-block
        # sums, `apply(...)` finishes (or is handed) the per-channel sums and writes dy.
        if tail == TAIL_HEAD:
            dl = grads[0].float().contiguous()
            ncls = hw2.shape[0]
            head_w, head_b = ctx.head_params
            (dwb, cb_hw), (dbb, cb_hb) = _grad_buffer(head_w, ctx.needs_input_grad[12]), _grad_buffer(head_b, ctx.needs_input_grad[13])
            direct = cb_hw is not None and cb_hb is not None and dwb.stride(0) == Cout and dwb.stride(1) == 1 and \
                dbb.is_contiguous() and dwb.dtype == torch.float32 and dbb.dtype == torch.float32
            dhw = dwb if direct else torch.empty((ncls, Cout), dtype=torch.float32, device=dev)
            dhb = dbb if direct else torch.empty(ncls, dtype=torch.float32, device=dev)
            nbytes = LIB.query("uh_bn_relu_head_bwd_ws_bytes", n, Cout, ncls)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

            def reduce():
                LIB.call("uh_bn_relu_head_bwd_reduce", dl.data_ptr(), hw2.data_ptr(), *bn_args, partials.data_ptr(),
                         dhw.data_ptr(), dhb.data_ptr(), ws.data_ptr(), nbytes, n, Cout, ncls, dt, _stream())
                if direct:
                    cb_hw()
                    cb_hb()

            def apply(part_ptr, nb, dg_ptr, db_ptr, n_total):
                LIB.call("uh_bn_relu_head_bwd_apply", dl.data_ptr(), hw2.data_ptr(), *bn_args, part_ptr, nb, dg_ptr, db_ptr,
                         dy.data_ptr(), Cout, n, n_total, Cout, ncls, dt, _stream())
            dhead = (None, None) if direct else (dhw.view(head_w.shape) if ctx.needs_input_grad[12] else None,
                                                 dhb if ctx.needs_input_grad[13] else None)
        elif tail == TAIL_POOL and grads[1] is not None:
            dskip, dpool = grads
            dpool = dense_nhwc(dpool if dpool.dtype == y.dtype else dpool.to(y.dtype))
            if dskip is not None:
                dskip = dense_nhwc(dskip if dskip.dtype == y.dtype else dskip.to(y.dtype))
            sk = (_p(dskip), 0 if dskip is None else pixel_ld(dskip), dpool.data_ptr(), pixel_ld(dpool))

            def reduce():
                LIB.call("uh_bn_relu_pool_bwd_reduce", *sk, *bn_args, partials.data_ptr(), B, H, W, Cout, dt, _stream())

            def apply(part_ptr, nb, dg_ptr, db_ptr, n_total):
                LIB.call("uh_bn_relu_pool_bwd_apply", *sk, *bn_args, part_ptr, nb, dg_ptr, db_ptr, dy.data_ptr(), Cout,
                         B, H, W, n_total, Cout, dt, _stream())
        else:
            dz = grads[0]
            if dz is None:          # pool tail whose outputs were both unused
                return (None,) * 19
            dz = dense_nhwc(dz if dz.dtype == y.dtype else dz.to(y.dtype))
            if tail == TAIL_UP:     # the gradient arrives for the up-sampled tensor: transpose of the interpolation first
                Ho, Wo, pt, pl = ctx.up
                du, dz = dz, torch.empty((B, H, W, Cout), dtype=y.dtype, device=dev)
                LIB.call("uh_upsample2x_bwd", du.data_ptr(), pixel_ld(du), dz.data_ptr(), Cout, B, H, W, Cout, Ho, Wo, pt, pl,
                         dt, _stream())
            link = ctx.bnsum_pub
            if link is not None:
                # the sums came with the gradient (uh_conv3x3_dgrad_bnsum in the consumer's backward) -- if this IS that gradient
                # (same storage, same extent, and not written since the sums were formed: a tensor hook or an in-place
                # accumulation that edits the gradient keeps the storage and bumps the version counter)
                if link.partials is not None and link.dz is not None and dz.data_ptr() == link.dz.data_ptr() and \
                        dz.shape == link.dz.shape and dz.stride() == link.dz.stride() and dz._version == link.dz_version:
                    partials, nblk = link.partials, link.rows
                else:
                    link = None
                ctx.bnsum_pub.partials = ctx.bnsum_pub.dz = ctx.bnsum_pub.y = ctx.bnsum_pub.coef = None
            fused_sums = link is not None

            def reduce():
                if not fused_sums:
                    LIB.call("uh_bn_relu_bwd_reduce", dz.data_ptr(), pixel_ld(dz), *bn_args, partials.data_ptr(), n, Cout, dt,
                             _stream())

            def apply(part_ptr, nb, dg_ptr, db_ptr, n_total):
                LIB.call("uh_bn_relu_bwd_apply", dz.data_ptr(), pixel_ld(dz), *bn_args, part_ptr, nb, dg_ptr, db_ptr,
                         dy.data_ptr(), Cout, n, n_total, Cout, dt, _stream())
        reduce()
        gamma_p, beta_p = ctx.bn_params
        (dgamma, cb_g), (dbeta, cb_b) = _grad_buffer(gamma_p, ctx.needs_input_grad[3]), _grad_buffer(beta_p, ctx.needs_input_grad[4])
        dy = torch.empty_like(y)
        if ctx.sync_bn is None:
            apply(partials.data_ptr(), nblk, dgamma.data_ptr(), dbeta.data_ptr(), 0)
        else:
            # SyncBN: the parameter gradients stay LOCAL sums (the gradient all-reduce SUMS them like every other
            # gradient: the loss is already normalised by the global batch); the dx formula needs the GLOBAL sums and the
            # global pixel count
            import torch.distributed as dist
            LIB.call("uh_bn_bwd_finalize", partials.data_ptr(), nblk, Cout, dgamma.data_ptr(), dbeta.data_ptr(), _stream())
            glob = torch.cat([dgamma.reshape(-1), dbeta.reshape(-1)])
            dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=ctx.sync_bn[0])
            apply(None, 0, glob[:Cout].data_ptr(), glob[Cout:].data_ptr(), ctx.n_total)
        # backward-data first: it is the only consumer on the critical path (the next layer's BatchNorm backward waits
        # for it).  Backward-weights then goes to the side stream BEHIND it, so that it runs beside the HBM-bound
        # kernels of the layers that follow (BatchNorm backward, pool / upsample backward) instead of beside this
        # layer's own MFMA-bound backward-data.
        dx0 = dx1 = None
        if wd is not None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            use = ctx.bnsum_use
            rows = 0
            if use is not None and use.y is not None and use.y.shape == (B, H, W, Cin) and use.y.is_contiguous() and \
                    (H, Cin) not in BNSUM_OFF:
                rows = LIB.query("uh_conv3x3_dgrad_bnsum_rows", B, H, W, Cout, Cin, Cout, Cin, Cin, ctx.cdt)
            if rows > 0:
                # backward-data + the BatchNorm-backward sums of the layer in front (its reduce pass is not launched)
                dx = torch.empty((B, H, W, Cin), dtype=dy.dtype, device=dev)
                bsum = torch.empty(rows * 2 * Cin, dtype=torch.float32, device=dev)
                # (a kernel family of its own in bench.py's profile: the same MFMA work as plain backward-data plus a read of
                # `use.y` and the sums -- its time is not comparable with the plain launches')
                with _Timed("conv3x3_dgrad_bnsum_mfma", 2.0 * B * H * W * Cin * 9 * Cout, ("dgrad", B, H, W, Cout, Cin)):
                    LIB.call("uh_conv3x3_dgrad_bnsum", dy.data_ptr(), Cout, Cout, wd.data_ptr(), dx.data_ptr(), Cin, Cin,
                             use.y.data_ptr(), Cin, use.coef.data_ptr(), bsum.data_ptr(), B, H, W,
                             ctx.cdt | (UH_WFRAG if ctx.frag_d else 0), _stream())
                use.partials, use.rows, use.dz, use.dz_version = bsum, rows, dx, dx._version
            else:
                dx, _, _ = conv3x3_fwd(dy, None, wd, Cin, False, ctx.cdt, ctx.frag_d)
            dx0 = dx[..., :C0] if ctx.needs_input_grad[0] else None
            dx1 = dx[..., C0:] if (x1 is not None and ctx.needs_input_grad[1]) else None
        # weight gradient: straight into the parameter's layout when that IS KRSC (channels_last weights)
        dweight = None
        side_done = None

        def run_wgrad(out_, defer_cb=None):
            if pre_coef is None:
                return conv3x3_wgrad(dy, x0, x1, out_, ctx.cdt == UH_F32X3, defer_cb)
            conv3x3_wgrad_pre(dy, x0, pre_coef, out_)
            return False

        if ctx.needs_input_grad[2]:
            dweight, cb_w = _grad_buffer(weight)
            if cb_w is not None and WGRAD_STREAM is not None and _is_krsc_dense(weight):
                side = WGRAD_STREAM
                ev = torch.cuda.Event()
                ev.record()                         # dy (and this layer's backward-data) are enqueued before this point
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    run_wgrad(dweight)
                    side_done = torch.cuda.Event()
                    side_done.record()              # the gradient all-reduce of this parameter's bucket waits for THIS
                for t_ in (dy, x0, x1, pre_coef):
                    if t_ is not None:
                        t_.record_stream(side)      # the caching allocator must not recycle them under the side stream
            elif _is_krsc_dense(weight):
                # (straight into the optimizer's flat buffer: the closing reduction may wait for the batched launch, and the
                # "gradient ready" callback with it)
                if run_wgrad(dweight, cb_w):
                    cb_w = None
                    dweight = None
            else:
                dwk = torch.empty(Cout * 9 * Cin, dtype=torch.float32, device=dev)
                run_wgrad(dwk)
                sO, sI, sH, sW = dweight.stride()
                LIB.call("uh_unpack_dw3x3", dwk.data_ptr(), dweight.data_ptr(), sO, sI, sH, sW, Cout, Cin, _stream())
            if cb_w is not None:
                cb_w(side_done)
                dweight = None
        if cb_g is not None:
            cb_g()
            dgamma = None
        if cb_b is not None:
            cb_b()
            dbeta = None
        if not ctx.needs_input_grad[3]:
            dgamma = None
        if not ctx.needs_input_grad[4]:
            dbeta = None
        return dx0, dx1, dweight, dgamma, dbeta, None, None, None, None, None, None, None, dhead[0], dhead[1], None, None, None, None, None


# ----------------------------------------------------------------------------- the stem, output recomputed
# The first layer of the network (inc.double_conv.0-2: Conv2d(Cin <= 4 -> 64) -> BatchNorm2d -> ReLU) with its conv output
# RECOMPUTED by every consumer instead of stored (csrc/conv3x3.hip, stem_bn_bwd_v3): 9 multiply-adds per element against four HBM
# passes over the largest tensor of the model.  bf16 training, single-channel images by default (STEM_RECOMPUTE_MAX_CIN).
STEM_RECOMPUTE = os.environ.get("UH_STEM_RECOMPUTE", "1") != "0"
STEM_RECOMPUTE_MAX_CIN = 1


def stem_recompute_ok(x0: torch.Tensor, Cin: int, Cout: int) -> bool:
    return bool(STEM_RECOMPUTE and x0.dtype == torch.bfloat16 and Cin <= STEM_RECOMPUTE_MAX_CIN and x0.shape[-1] == Cin and
                LIB.query("uh_stem_ok", Cin, Cout, UH_BF16) and x0.shape[0] * x0.shape[1] * x0.shape[2] < 2 ** 31)


def _bn_train_coefficients(stats, nslab, Cout, n, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps):
    """Per-channel [scale | shift | mean | rstd] from the conv kernels' statistics rows (+ running statistics /
    num_batches_tracked update, + the cross-rank merge under SyncBN).  -> (coef, n_total)"""
    global BN_STATS_EPOCH
    BN_STATS_EPOCH += 1
    dev = stats.device
    coef = torch.empty(4 * Cout, dtype=torch.float32, device=dev)
    scale, shift, mean, rstd = coef[:Cout], coef[Cout:2 * Cout], coef[2 * Cout:3 * Cout], coef[3 * Cout:]
    g32 = gamma if gamma.dtype == torch.float32 else gamma.float()
    b32 = beta if beta.dtype == torch.float32 else beta.float()
    nbt = num_batches_tracked
    fused_nbt = nbt is not None and nbt.is_cuda and nbt.dtype == torch.int64
    nbt_ptr = nbt.data_ptr() if fused_nbt else None
    n_total = n
    if SYNC_BN is None:
        LIB.call("uh_bn_finalize", stats.data_ptr(), nslab, Cout, n, g32.data_ptr(), b32.data_ptr(), _p(running_mean),
                 _p(running_var), nbt_ptr, float(momentum), float(eps), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                 rstd.data_ptr(), None, _stream())
    else:
        m2 = torch.empty(Cout, dtype=torch.float32, device=dev)
        LIB.call("uh_bn_finalize", stats.data_ptr(), nslab, Cout, n, g32.data_ptr(), b32.data_ptr(), None, None, None,
                 float(momentum), float(eps), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                 m2.data_ptr(), _stream())
        n_total = _sync_bn_forward(coef, m2, n, Cout, g32, b32, running_mean, running_var, nbt_ptr, momentum, eps)
    if nbt is not None and not fused_nbt:
        nbt.add_(1)
    return coef, n_total


class StemConvBnReluFn(Function):
    """(Conv2d(1 -> 64, 3x3, pad 1, no bias) -> BatchNorm2d -> ReLU) of unet_parts.py:15-17 for the network's first
    layer in training: the conv output is never written; statistics, the activation, the BatchNorm-backward sums and the
    filter gradient are each computed from the IMAGE (uh_stem_*).  The image gets no gradient."""

    @staticmethod
    def forward(ctx, x0, weight, gamma, beta, running_mean, running_var, num_batches_tracked, momentum: float, eps: float):
        _require_gpu(x0, "image")
        x0 = dense_nhwc(x0)
        B, H, W, Cin = x0.shape
        Cout = weight.shape[0]
        if weight.shape[1] != Cin or not stem_recompute_ok(x0, Cin, Cout):
            raise RuntimeError("StemConvBnReluFn: bf16 single-channel image into 64 output channels only")
        dev = x0.device
        hit = WEIGHT_PACK.lookup(weight, x0.dtype, False, False) if WEIGHT_PACK is not None else None
        wf = hit[0] if hit is not None else pack_w3x3(weight, x0.dtype, False)[0]
        n = B * H * W
        nslab = LIB.query("uh_conv3x3_stat_slabs", B, H, W, Cin, Cout, UH_BF16)
        stats = torch.empty(nslab * (2 * Cout + 2), dtype=torch.float32, device=dev)
        with _Timed("conv3x3_fwd_stem", 2.0 * n * Cout * 9 * Cin):
            LIB.call("uh_stem_stats", x0.data_ptr(), Cin, pixel_ld(x0), wf.data_ptr(), stats.data_ptr(), B, H, W, UH_BF16, _stream())
        coef, n_total = _bn_train_coefficients(stats, nslab, Cout, n, gamma, beta, running_mean, running_var,
                                               num_batches_tracked, momentum, eps)
        z = torch.empty((B, H, W, Cout), dtype=x0.dtype, device=dev)
        with _Timed("conv3x3_fwd_stem", 2.0 * n * Cout * 9 * Cin):
            LIB.call("uh_stem_bn_relu_fwd", x0.data_ptr(), Cin, pixel_ld(x0), wf.data_ptr(), coef.data_ptr(), coef[Cout:].data_ptr(),
                     z.data_ptr(), Cout, B, H, W, UH_BF16, _stream())
        ctx.save_for_backward(x0, coef, wf, weight)
        ctx.bn_params = (gamma, beta)
        ctx.dims = (B, H, W, Cin, Cout)
        ctx.n_total = n_total
        ctx.sync_bn = SYNC_BN                    # (None = per-rank statistics; a one-rank group under UH_DP_FORCE_SYNC still runs the collectives)
        return z

    @staticmethod
    def backward(ctx, dz):
        x0, coef, wf, weight = ctx.saved_tensors
        B, H, W, Cin, Cout = ctx.dims
        dev = x0.device
        dz = dense_nhwc(dz if dz.dtype == x0.dtype else dz.to(x0.dtype))
        scale, shift, mean, rstd = coef[:Cout], coef[Cout:2 * Cout], coef[2 * Cout:3 * Cout], coef[3 * Cout:]
        args = (dz.data_ptr(), pixel_ld(dz), x0.data_ptr(), Cin, pixel_ld(x0), wf.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                mean.data_ptr(), rstd.data_ptr())
        nblk = LIB.query("uh_stem_nblk", B, H, W)
        partials = torch.empty(nblk * 2 * Cout, dtype=torch.float32, device=dev)
        LIB.call("uh_stem_bn_relu_bwd_reduce", *args, partials.data_ptr(), B, H, W, UH_BF16, _stream())
        gamma_p, beta_p = ctx.bn_params
        (dgamma, cb_g), (dbeta, cb_b) = _grad_buffer(gamma_p, ctx.needs_input_grad[2]), _grad_buffer(beta_p, ctx.needs_input_grad[3])
        LIB.call("uh_bn_bwd_finalize", partials.data_ptr(), nblk, Cout, dgamma.data_ptr(), dbeta.data_ptr(), _stream())
        sums_g, sums_b = dgamma, dbeta
        if ctx.sync_bn is not None:             # the dy formula needs the GLOBAL sums; the parameter gradients stay local
            import torch.distributed as dist
            glob = torch.cat([dgamma.reshape(-1), dbeta.reshape(-1)])
            dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=ctx.sync_bn[0])
            sums_g, sums_b = glob[:Cout], glob[Cout:]
        dweight = None
        if ctx.needs_input_grad[1]:
            dweight, cb_w = _grad_buffer(weight)
            nbytes = LIB.query("uh_stem_bwd_wgrad_ws_bytes", B, H, W, Cin)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            direct = _is_krsc_dense(weight)
            dwk = dweight if direct else torch.empty(Cout * 9 * Cin, dtype=torch.float32, device=dev)
            with _Timed("conv3x3_wgrad_stem", 2.0 * B * H * W * Cout * 9 * Cin):
                LIB.call("uh_stem_bn_relu_bwd_wgrad", *args, sums_g.data_ptr(), sums_b.data_ptr(), ctx.n_total, dwk.data_ptr(),
                         ws.data_ptr(), nbytes, B, H, W, UH_BF16, _stream())
            if not direct:
                sO, sI, sH, sW = dweight.stride()
                LIB.call("uh_unpack_dw3x3", dwk.data_ptr(), dweight.data_ptr(), sO, sI, sH, sW, Cout, Cin, _stream())
            if cb_w is not None:
                cb_w(None)
                dweight = None
        if cb_g is not None:
            cb_g()
            dgamma = None
        if cb_b is not None:
            cb_b()
            dbeta = None
        if not ctx.needs_input_grad[2]:
            dgamma = None
        if not ctx.needs_input_grad[3]:
            dbeta = None
        return None, dweight, dgamma, dbeta, None, None, None, None, None


# ----------------------------------------------------------------------------- small-width conv + BN + ReLU
# Small-width layers keep their activations at the real channel count in HBM (False: 64-channel zero-padded tensors, the
# first implementation -- kept for A/B measurements and as the path for channel counts that are not 16-byte multiples).
NARROW_IO = True


def _rup64(c: int) -> int:
    return (c + 63) // 64 * 64


def narrow_ok(x0: torch.Tensor, x1: Optional[torch.Tensor], Cout: int) -> bool:
    """Can this layer run on the narrow-tensor conv entry points?  (every stored channel count a multiple of one
    16-byte piece, 16-byte aligned pixel rows)"""
    vec = 16 // x0.element_size()
    for t in (x0, x1):
        if t is None:
            continue
        if t.shape[-1] % vec or (pixel_ld(t) * t.element_size()) % 16 or t.data_ptr() % 16:
            return False
    return Cout % vec == 0


class ConvBnReluNarrowFn(Function):
    """(Conv2d 3x3 -> BatchNorm2d -> ReLU) of unet_parts.py:15-20 for the small-width models (UNet_S / UNet_T,
    unet_model.py:52-126): COMPUTED as the next 64-aligned layer -- zero filters / unit gamma in the padding, so the
    MFMA kernels of the full-width UNet apply -- while every activation in HBM (x, raw conv output, z and their
    gradients) keeps its real channel count: the conv kernels read channels beyond the stored count as zeros and never
    write them (uh_conv3x3_fwd_narrow / uh_conv3x3_wgrad_narrow).  `c0_true`: x0 may itself carry zero channels up to
    a 16-byte piece (the 1- or 3-channel input image); only its first c0_true channels meet filter taps."""

    @staticmethod
    def forward(ctx, x0, x1, weight, gamma, beta, running_mean, running_var, num_batches_tracked,
                training: bool, momentum: float, eps: float, c0_true: int):
        _require_gpu(x0, "activation")
        x0 = dense_nhwc(x0)
        x1 = None if x1 is None else dense_nhwc(x1)
        B, H, W, C0m = x0.shape
        C1m = 0 if x1 is None else x1.shape[3]
        Cout, Cin = weight.shape[0], weight.shape[1]
        if Cin != c0_true + C1m or c0_true > C0m:
            raise RuntimeError(f"conv expects {Cin} input channels, got {c0_true}+{C1m}")
        if not narrow_ok(x0, x1, Cout):
            raise RuntimeError("narrow conv: stored channel counts / strides must be multiples of 16 bytes")
        Cp0, Cp1, Cop = _rup64(C0m), _rup64(C1m) if C1m else 0, _rup64(Cout)
        Cinp = Cp0 + Cp1
        dev = x0.device
        need_dx = (ctx.needs_input_grad[0], x1 is not None and ctx.needs_input_grad[1])
        cdt = conv_dt(x0, Cp0, Cp1, Cop, any(need_dx) and training)
        single = x1 is None
        want_wd = (training and need_dx[0], training and need_dx[1])
        w32 = weight if weight.dtype == torch.float32 else weight.float()
        wd0 = wd1 = None
        if cdt == UH_F32X3:
            # bf16x3 packs are [hi | lo] pairs, so the per-source backward-data filters cannot be row blocks of one pack:
            # pad with torch, pack per source
            with torch.no_grad():
                wp = torch.zeros(Cop, Cinp, 3, 3, dtype=torch.float32, device=dev)
                wp[:Cout, :c0_true] = w32[:, :c0_true]
                if C1m:
                    wp[:Cout, Cp0:Cp0 + C1m] = w32[:, c0_true:]
            wf, wd0 = pack_w3x3(wp, x0.dtype, want_wd[0] and single, cdt)
            if training and not single:
                wd0 = pack_w3x3(wp[:, :Cp0], x0.dtype, True, cdt)[1] if want_wd[0] else None
                wd1 = pack_w3x3(wp[:, Cp0:], x0.dtype, True, cdt)[1] if want_wd[1] else None
        else:
            # one launch: zero-padded forward pack + backward-data pack whose row blocks are the per-source filters
            wf = torch.empty(Cop * 9 * Cinp, dtype=x0.dtype, device=dev)
            wd = torch.empty(Cop * 9 * Cinp, dtype=x0.dtype, device=dev) if any(want_wd) else None
            sO, sI, sH, sW = w32.stride()
            LIB.call("uh_pack_w3x3_padded", w32.data_ptr(), sO, sI, sH, sW, Cout, c0_true, C1m, Cop, Cp0, Cp1, wf.data_ptr(),
                     _p(wd), cdt, _stream())
            if wd is not None:
                wd0 = wd[:Cp0 * 9 * Cop] if want_wd[0] else None
                wd1 = wd[Cp0 * 9 * Cop:] if want_wd[1] else None
        g32 = gamma if gamma.dtype == torch.float32 else gamma.float()
        b32 = beta if beta.dtype == torch.float32 else beta.float()
        n = B * H * W
        ld0, ld1 = pixel_ld(x0), 0 if x1 is None else pixel_ld(x1)
        flops = 2.0 * n * Cop * 9 * Cinp
        if not training:
            # scale / shift are read in 16-byte pieces for every padded channel group: Cop entries, the first Cout real
            scale, shift = bn_eval_coeffs_cached(gamma, beta, running_mean, running_var, eps, Cout, Cop)
            z = torch.empty(B, H, W, Cout, dtype=x0.dtype, device=dev)
            with _Timed("conv3x3_fwd_narrow", flops):
                LIB.call("uh_conv3x3_fwd_narrow", x0.data_ptr(), Cp0, C0m, ld0, _p(x1), Cp1, C1m, ld1, wf.data_ptr(),
                         z.data_ptr(), Cout, Cop, Cout, None, scale.data_ptr(), shift.data_ptr(), B, H, W, cdt, _stream())
            ctx.training = False
            return z
        coef = torch.empty(4 * Cout, dtype=torch.float32, device=dev)
        scale, shift, mean, rstd = coef[:Cout], coef[Cout:2 * Cout], coef[2 * Cout:3 * Cout], coef[3 * Cout:]
        global BN_STATS_EPOCH
        BN_STATS_EPOCH += 1                       # running statistics are about to change under torch's feet
        y = torch.empty(B, H, W, Cout, dtype=x0.dtype, device=dev)
        nslab = LIB.query("uh_conv3x3_stat_slabs", B, H, W, Cinp, Cop, cdt)
        stats = torch.empty(nslab * (2 * Cop + 2), dtype=torch.float32, device=dev)
        with _Timed("conv3x3_fwd_narrow", flops):
            LIB.call("uh_conv3x3_fwd_narrow", x0.data_ptr(), Cp0, C0m, ld0, _p(x1), Cp1, C1m, ld1, wf.data_ptr(),
                     y.data_ptr(), Cout, Cop, Cout, stats.data_ptr(), None, None, B, H, W, cdt, _stream())
        nbt = num_batches_tracked
        fused_nbt = nbt is not None and nbt.is_cuda and nbt.dtype == torch.int64
        nbt_ptr = nbt.data_ptr() if fused_nbt else None
        n_total = n
        # statistics rows are Cop channels wide (the conv is computed for the padded layer); only the Cout real ones count
        if SYNC_BN is None:
            LIB.call("uh_bn_finalize_ld", stats.data_ptr(), nslab, Cop, Cout, n, g32.data_ptr(), b32.data_ptr(),
                     _p(running_mean), _p(running_var), nbt_ptr, float(momentum), float(eps), scale.data_ptr(),
                     shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, _stream())
        else:
            m2 = torch.empty(Cout, dtype=torch.float32, device=dev)
            LIB.call("uh_bn_finalize_ld", stats.data_ptr(), nslab, Cop, Cout, n, g32.data_ptr(), b32.data_ptr(), None, None,
                     None, float(momentum), float(eps), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                     m2.data_ptr(), _stream())
            n_total = _sync_bn_forward(coef, m2, n, Cout, g32, b32, running_mean, running_var, nbt_ptr, momentum, eps)
        if nbt is not None and not fused_nbt:
            nbt.add_(1)
        z = torch.empty_like(y)
        LIB.call("uh_bn_relu_apply", y.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(), z.data_ptr(), Cout, n, Cout,
                 _dt(y), _stream())
        ctx.save_for_backward(x0, x1, y, coef, wd0, wd1, weight)
        ctx.bn_params = (gamma, beta)
        ctx.training = True
        ctx.dims = (B, H, W, C0m, C1m, Cout, c0_true)
        ctx.cdt = cdt
        ctx.n_total = n_total
        ctx.sync_bn = SYNC_BN                    # (None = per-rank statistics; a one-rank group under UH_DP_FORCE_SYNC still runs the collectives)
        return z

    @staticmethod
    def backward(ctx, dz):
        if not ctx.training:
            raise RuntimeError("backward through eval-mode BatchNorm is not part of the train path")
        x0, x1, y, coef, wd0, wd1, weight = ctx.saved_tensors
        B, H, W, C0m, C1m, Cout, c0_true = ctx.dims
        Cp0, Cp1, Cop = _rup64(C0m), _rup64(C1m) if C1m else 0, _rup64(Cout)
        Cinp = Cp0 + Cp1
        n = B * H * W
        dev = y.device
        dz = dense_nhwc(dz if dz.dtype == y.dtype else dz.to(y.dtype))
        scale, shift, mean, rstd = coef[:Cout], coef[Cout:2 * Cout], coef[2 * Cout:3 * Cout], coef[3 * Cout:]
        dt = _dt(y)
        nblk = LIB.query("uh_bn_bwd_nblk", n, Cout)
        partials = torch.empty(nblk * 2 * Cout, dtype=torch.float32, device=dev)
        LIB.call("uh_bn_relu_bwd_reduce", dz.data_ptr(), pixel_ld(dz), y.data_ptr(), Cout, scale.data_ptr(),
                 shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), partials.data_ptr(), n, Cout, dt, _stream())
        gamma_p, beta_p = ctx.bn_params
        (dgamma, cb_g), (dbeta, cb_b) = _grad_buffer(gamma_p, ctx.needs_input_grad[3]), _grad_buffer(beta_p, ctx.needs_input_grad[4])
        dy = torch.empty_like(y)
        if ctx.sync_bn is None:
            LIB.call("uh_bn_relu_bwd_apply", dz.data_ptr(), pixel_ld(dz), y.data_ptr(), Cout, scale.data_ptr(),
                     shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), partials.data_ptr(), nblk, dgamma.data_ptr(),
                     dbeta.data_ptr(), dy.data_ptr(), Cout, n, 0, Cout, dt, _stream())
        else:
            import torch.distributed as dist
            LIB.call("uh_bn_bwd_finalize", partials.data_ptr(), nblk, Cout, dgamma.data_ptr(), dbeta.data_ptr(), _stream())
            glob = torch.cat([dgamma.reshape(-1), dbeta.reshape(-1)])
            dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=ctx.sync_bn[0])
            LIB.call("uh_bn_relu_bwd_apply", dz.data_ptr(), pixel_ld(dz), y.data_ptr(), Cout, scale.data_ptr(),
                     shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, 0, glob[:Cout].data_ptr(),
                     glob[Cout:].data_ptr(), dy.data_ptr(), Cout, n, ctx.n_total, Cout, dt, _stream())
        # backward-data: the padded layer's transposed conv, once per source, each writing its own narrow dx
        dx = [None, None]
        for i, (wd, Cp, Cm) in enumerate(((wd0, Cp0, C0m), (wd1, Cp1, C1m))):
            if wd is None or not ctx.needs_input_grad[i]:
                continue
            dx[i] = torch.empty(B, H, W, Cm, dtype=y.dtype, device=dev)
            with _Timed("conv3x3_fwd_narrow", 2.0 * n * Cop * 9 * Cp):
                LIB.call("uh_conv3x3_fwd_narrow", dy.data_ptr(), Cop, Cout, Cout, None, 0, 0, 0, wd.data_ptr(),
                         dx[i].data_ptr(), Cm, Cp, Cm, None, None, None, B, H, W, ctx.cdt, _stream())
        dweight = None
        if ctx.needs_input_grad[2]:
            wdt = UH_F32X3 if (ctx.cdt == UH_F32X3) else dt
            nbytes = LIB.query("uh_conv3x3_wgrad_ws_bytes", B, H, W, Cinp, Cop, dt)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            dwk = torch.empty(Cop, 3, 3, Cinp, dtype=torch.float32, device=dev)
            with _Timed("conv3x3_wgrad_narrow", 2.0 * n * Cop * 9 * Cinp):
                LIB.call("uh_conv3x3_wgrad_narrow", dy.data_ptr(), Cout, Cop, Cout, x0.data_ptr(), Cp0, C0m, pixel_ld(x0),
                         _p(x1), Cp1, C1m, 0 if x1 is None else pixel_ld(x1), dwk.data_ptr(), ws.data_ptr(), nbytes,
                         B, H, W, wdt, _stream())
            dweight, cb_w = _grad_buffer(weight)
            dweight[:, :c0_true].copy_(dwk[:Cout, :, :, :c0_true].permute(0, 3, 1, 2))
            if C1m:
                dweight[:, c0_true:].copy_(dwk[:Cout, :, :, Cp0:Cp0 + C1m].permute(0, 3, 1, 2))
            if cb_w is not None:
                cb_w()
                dweight = None
        if cb_g is not None:
            cb_g()
            dgamma = None
        if cb_b is not None:
            cb_b()
            dbeta = None
        if not ctx.needs_input_grad[3]:
            dgamma = None
        if not ctx.needs_input_grad[4]:
            dbeta = None
        return dx[0], dx[1], dweight, dgamma, dbeta, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------- max-pool
def _maxpool_fwd(x):
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    LIB.call("uh_maxpool2_fwd", x.data_ptr(), pixel_ld(x), y.data_ptr(), C, B, H, W, C, _dt(x), _stream())
    return y


def _maxpool_bwd(x, dy, dskip):
    B, H, W, C = x.shape
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    dy = dense_nhwc(dy if dy.dtype == x.dtype else dy.to(x.dtype))
    if dskip is not None:
        dskip = dense_nhwc(dskip if dskip.dtype == x.dtype else dskip.to(x.dtype))
    LIB.call("uh_maxpool2_bwd", x.data_ptr(), pixel_ld(x), dy.data_ptr(), pixel_ld(dy), _p(dskip),
             0 if dskip is None else pixel_ld(dskip), dx.data_ptr(), C, B, H, W, C, _dt(x), _stream())
    return dx


class MaxPool2Fn(Function):
    """nn.MaxPool2d(2) (unet_parts.py:32)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, "activation")
        x = dense_nhwc(x)
        ctx.save_for_backward(x)
        return _maxpool_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _maxpool_bwd(x, dy, None)


class PoolSplitFn(Function):
    """x -> (x as the skip connection, maxpool(x)).  One node so that the backward adds the skip
    gradient and routes the pooled gradient in a single pass (unet_model.py:28-32 uses every
    encoder output twice)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, "activation")
        x = dense_nhwc(x)
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return x.view_as(x), _maxpool_fwd(x)

    @staticmethod
    def backward(ctx, dskip, dpool):
        (x,) = ctx.saved_tensors
        if dpool is None:
            return dskip
        return _maxpool_bwd(x, dpool, dskip)


# ----------------------------------------------------------------------------- upsample / convT (+ pad)
def _pad_geometry(h, w, Ho, Wo):
    """F.pad(x1, [dX//2, dX-dX//2, dY//2, dY-dY//2]) of unet_parts.py:85-88 (Python floor division)."""
    dY, dX = Ho - 2 * h, Wo - 2 * w
    return dY // 2, dX // 2


class UpsampleBilinearPadFn(Function):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) + F.pad (unet_parts.py:70,85-88)."""

    @staticmethod
    def forward(ctx, x, Ho: int, Wo: int):
        _require_gpu(x, "activation")
        x = dense_nhwc(x)
        B, h, w, C = x.shape
        pt, pl = _pad_geometry(h, w, Ho, Wo)
        y = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
        LIB.call("uh_upsample2x_fwd", x.data_ptr(), pixel_ld(x), y.data_ptr(), C, B, h, w, C, Ho, Wo, pt, pl,
                 _dt(x), _stream())
        ctx.geom = (B, h, w, C, Ho, Wo, pt, pl)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, h, w, C, Ho, Wo, pt, pl = ctx.geom
        dy = dense_nhwc(dy)
        dx = torch.empty((B, h, w, C), dtype=dy.dtype, device=dy.device)
        LIB.call("uh_upsample2x_bwd", dy.data_ptr(), pixel_ld(dy), dx.data_ptr(), C, B, h, w, C, Ho, Wo, pt, pl,
                 _dt(dy), _stream())
        return dx, None, None


class ConvTranspose2x2PadFn(Function):
    """nn.ConvTranspose2d(Cin, Cin//2, 2, stride=2) + F.pad (unet_parts.py:73,85-88).  MFMA GEMMs over pixels
    (csrc/convt_mfma.hip) when the shape qualifies, else the LDS-tiled SIMT kernels of csrc/convt_1x1.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, Ho: int, Wo: int):
        _require_gpu(x, "activation")
        x = dense_nhwc(x)
        B, h, w, Cin = x.shape
        Cout = weight.shape[1]
        wc = weight.contiguous().float()
        bc = bias.contiguous().float()
        pt, pl = _pad_geometry(h, w, Ho, Wo)
        y = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
        dt = _dt(x)
        mfma = bool(LIB.query("uh_convt2x2_mfma_ok", B, h, w, Cin, Cout, Ho, Wo, dt))
        if mfma and dt == UH_F32 and FP32_MODE == "bf16x3":
            dt = UH_F32X3                    # split products on the bf16 matrix pipe (both operands split in registers)
        wd = None
        if mfma:
            wf = torch.empty(Cin * Cout * 4, dtype=x.dtype, device=x.device)
            wd = torch.empty(Cin * Cout * 4, dtype=x.dtype, device=x.device)
            LIB.call("uh_convt2x2_pack", wc.data_ptr(), Cin, Cout, wf.data_ptr(), wd.data_ptr(), dt, _stream())
            with _Timed("convt2x2_fwd_mfma", 2.0 * B * h * w * Cin * 4 * Cout):
                LIB.call("uh_convt2x2_fwd_mfma", x.data_ptr(), pixel_ld(x), wf.data_ptr(), bc.data_ptr(), y.data_ptr(), Cout,
                         B, h, w, Cin, Cout, Ho, Wo, pt, pl, dt, _stream())
        else:
            LIB.call("uh_convt2x2_fwd", x.data_ptr(), pixel_ld(x), wc.data_ptr(), bc.data_ptr(), y.data_ptr(), Cout,
                     B, h, w, Cin, Cout, Ho, Wo, pt, pl, dt, _stream())
        ctx.save_for_backward(x, wc, wd)
        ctx.geom = (B, h, w, Cin, Cout, Ho, Wo, pt, pl)
        ctx.dt = dt
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wc, wd = ctx.saved_tensors
        B, h, w, Cin, Cout, Ho, Wo, pt, pl = ctx.geom
        dy = dense_nhwc(dy if dy.dtype == x.dtype else dy.to(x.dtype))
        dt = ctx.dt
        mfma = wd is not None
        # dy usually is the [.., C:] half of the gradient of the concatenated tensor (pixel stride 2 x Cout): the MFMA kernels address
        # it through a 2 GiB buffer window that uh_convt2x2_mfma_ok sized for the PACKED tensor -- pack it when the strided walk
        # would pass that window (32 x 512 x 512 x 64 bf16 channels inside a 128-channel gradient: bench.py's global-batch-32 leg)
        if mfma and dy.shape[0] * dy.shape[1] * dy.shape[2] * pixel_ld(dy) * dy.element_size() >= (1 << 31) - 4096:
            dy = dy.contiguous()
        flops = 2.0 * B * h * w * Cin * 4 * Cout
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, h, w, Cin), dtype=x.dtype, device=x.device)
            if mfma:
                with _Timed("convt2x2_dgrad_mfma", flops):
                    LIB.call("uh_convt2x2_dgrad_mfma", dy.data_ptr(), pixel_ld(dy), wd.data_ptr(), dx.data_ptr(), Cin, B, h, w,
                             Cin, Cout, Ho, Wo, pt, pl, dt, _stream())
            else:
                LIB.call("uh_convt2x2_dgrad", dy.data_ptr(), pixel_ld(dy), wc.data_ptr(), dx.data_ptr(), Cin, B, h, w,
                         Cin, Cout, Ho, Wo, pt, pl, dt, _stream())
        dw = torch.empty((Cin, Cout, 2, 2), dtype=torch.float32, device=x.device)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
        if mfma:
            nbytes = LIB.query("uh_convt2x2_wgrad_mfma_ws_bytes", B, h, w, Cin, Cout, dt)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            with _Timed("convt2x2_wgrad_mfma", flops):
                LIB.call("uh_convt2x2_wgrad_mfma", dy.data_ptr(), pixel_ld(dy), x.data_ptr(), pixel_ld(x), dw.data_ptr(),
                         db.data_ptr(), ws.data_ptr(), nbytes, B, h, w, Cin, Cout, Ho, Wo, pt, pl, dt, _stream())
        else:
            nbytes = LIB.query("uh_convt2x2_wgrad_ws_bytes", B, h, w, Cin, Cout)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            LIB.call("uh_convt2x2_wgrad", dy.data_ptr(), pixel_ld(dy), x.data_ptr(), pixel_ld(x), dw.data_ptr(),
                     db.data_ptr(), ws.data_ptr(), nbytes, B, h, w, Cin, Cout, Ho, Wo, pt, pl, dt, _stream())
        return dx, dw, db, None, None


# ----------------------------------------------------------------------------- OutConv
class OutConv1x1Fn(Function):
    """nn.Conv2d(Cin, n_classes, kernel_size=1) with bias (unet_parts.py:103); logits are fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, "activation")
        x = dense_nhwc(x)
        B, H, W, Cin = x.shape
        ncls = weight.shape[0]
        w2 = weight.reshape(ncls, Cin).contiguous().float()
        b2 = bias.contiguous().float()
        logits = torch.empty((B, H, W, ncls), dtype=torch.float32, device=x.device)
        LIB.call("uh_conv1x1_fwd", x.data_ptr(), pixel_ld(x), w2.data_ptr(), b2.data_ptr(), logits.data_ptr(),
                 B * H * W, Cin, ncls, _dt(x), _stream())
        ctx.save_for_backward(x, w2)
        ctx.wshape = tuple(weight.shape)
        ctx.params = (weight, bias)
        return logits

    @staticmethod
    def backward(ctx, dl):
        x, w2 = ctx.saved_tensors
        B, H, W, Cin = x.shape
        ncls = w2.shape[0]
        n = B * H * W
        dl = dl.float().contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, H, W, Cin), dtype=x.dtype, device=x.device)
            LIB.call("uh_conv1x1_dgrad", dl.data_ptr(), w2.data_ptr(), dx.data_ptr(), Cin, n, Cin, ncls, _dt(x), _stream())
        nbytes = LIB.query("uh_conv1x1_wgrad_ws_bytes", n, Cin, ncls)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        # straight into the optimizer's flat gradient buffer when the parameter is registered there (FusedRMSprop) and its
        # [n_classes][Cin] plane is dense -- a 1x1 filter has the same memory order in contiguous and channels_last layout
        weight, bias = ctx.params
        (dwb, cb_w), (dbb, cb_b) = _grad_buffer(weight, ctx.needs_input_grad[1]), _grad_buffer(bias, ctx.needs_input_grad[2])
        direct = cb_w is not None and cb_b is not None and dwb.stride(0) == Cin and dwb.stride(1) == 1 and \
            dbb.is_contiguous() and dwb.dtype == torch.float32 and dbb.dtype == torch.float32
        if direct:
            dw, db = dwb, dbb
        else:
            dw = torch.empty((ncls, Cin), dtype=torch.float32, device=x.device)
            db = torch.empty(ncls, dtype=torch.float32, device=x.device)
        LIB.call("uh_conv1x1_wgrad", dl.data_ptr(), x.data_ptr(), pixel_ld(x), dw.data_ptr(), db.data_ptr(),
                 ws.data_ptr(), nbytes, n, Cin, ncls, _dt(x), _stream())
        if direct:
            cb_w()
            cb_b()
            return dx, None, None
        return dx, dw.view(ctx.wshape) if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None


# ----------------------------------------------------------------------------- losses
def loss_workspace(device) -> torch.Tensor:
    return torch.empty(LIB.query("uh_loss_ws_bytes", 1), dtype=torch.uint8, device=device)


def boundary_loss_value(pred: torch.Tensor, pstride: int, bstride: int, target: torch.Tensor, B: int, H: int, W: int,
                        edge_width: int, edge_weight: float, smooth: float = 1e-6) -> torch.Tensor:
    _require_gpu(pred, "prediction")
    _require_gpu(target, "target")
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    ws = loss_workspace(pred.device)
    LIB.call("uh_boundary_loss", pred.data_ptr(), pstride, bstride, target.data_ptr(), B, H, W, int(edge_width),
             float(edge_weight), float(smooth), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    return out


# Single-process binary loss as three launches (uh_seg_loss_binary_fused) instead of six + glue; False: the separate calls
# (what data-parallel runs use anyway: their sums are all-reduced between the calls).  The two are bit-identical (tests).
FUSE_LOSS = os.environ.get("UH_FUSE_LOSS", "1") != "0"
_LOSS_WS = {}


def _fused_loss_workspace(device) -> torch.Tensor:
    """One workspace per device for the fused loss (its three launches run back to back on one stream; a second stream
    using the loss concurrently on the same device is not a supported pattern of the train step)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _LOSS_WS.get(key)
    if ws is None:
        ws = _LOSS_WS[key] = torch.empty(LIB.query("uh_seg_loss_fused_ws_bytes"), dtype=torch.uint8, device=device)
    return ws


class SegLossBinaryFn(Function):
    """train.py:119-134 in one node: t = mask // mask_div; BCEWithLogits(mean) + dice_loss(sigmoid) +
    w_boundary * boundary_loss(logits, t, 51, 15).  Returns (total, bce, dice, boundary, nan_flag) as 0-dim tensors; only
    `total` carries gradient (boundary_loss is constant w.r.t. the logits); nan_flag (float 0/1, or None when
    the separate kernels ran) is train.py:149's isnan(loss) without a kernel of its own.
    `reduce_sums` (optional callable) all-reduces the 4 partial sums across data-parallel ranks so the
    Dice ratio and the BCE mean are those of the GLOBAL batch."""

    @staticmethod
    def forward(ctx, logits, mask, mask_div: int, w_boundary: float, edge_width: int, edge_weight: float,
                reduce_sums=None, world: int = 1):
        _require_gpu(logits, "logits")
        lg = logits.float().contiguous()
        if mask.dim() != 3 or lg.numel() != mask.numel():
            raise RuntimeError(f"binary seg loss expects mask [B,H,W] matching the logits, got {tuple(mask.shape)} "
                               f"vs {tuple(logits.shape)}")
        B, H, W = mask.shape
        n = lg.numel()
        mk = mask.contiguous()
        if mk.dtype != torch.int64:
            mk = mk.long()
        dev = lg.device
        sums = torch.empty(4, dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        fused = FUSE_LOSS and reduce_sums is None
        if fused:
            ws = _fused_loss_workspace(dev)
            LIB.call("uh_seg_loss_binary_fused", lg.data_ptr(), mk.data_ptr(), int(mask_div), B, H, W, int(edge_width),
                     float(edge_weight), 1e-6, float(w_boundary), float(round(n * world)), sums.data_ptr(), out.data_ptr(),
                     ws.data_ptr(), ws.numel(), _stream())
        else:
            ws = loss_workspace(dev)
            LIB.call("uh_bce_dice_sums", lg.data_ptr(), mk.data_ptr(), int(mask_div), None, n, sums.data_ptr(),
                     ws.data_ptr(), ws.numel(), _stream())
            if reduce_sums is not None:
                reduce_sums(sums)
            bl = None
            if w_boundary != 0.0:
                # boundary_loss(logits, (mask // mask_div).float(), 51, 15), train.py:134 -- the kernel divides the int64 mask itself
                bl = torch.empty(1, dtype=torch.float32, device=dev)
                LIB.call("uh_boundary_loss_mask", lg.data_ptr(), 1, H * W, mk.data_ptr(), int(mask_div), B, H, W, int(edge_width),
                         float(edge_weight), 1e-6, bl.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
            LIB.call("uh_seg_loss_binary_finish", sums.data_ptr(), float(round(n * world)), _p(bl), float(w_boundary),
                     out.data_ptr(), _stream())
        ctx.save_for_backward(lg, mk, sums)
        ctx.meta = (int(mask_div), n, world, logits.shape, logits.dtype)
        ctx.set_materialize_grads(False)
        total, bce, dice, bnd = out[0], out[1], out[2], out[3]
        nan_flag = out[4:5] if fused else None
        ctx.mark_non_differentiable(bce, dice, bnd)
        if nan_flag is not None:
            ctx.mark_non_differentiable(nan_flag)
        return total, bce, dice, bnd, nan_flag

    @staticmethod
    def backward(ctx, g_total, *_unused):
        lg, mk, sums = ctx.saved_tensors
        mask_div, n, world, shape, dtype = ctx.meta
        if g_total is None:
            return (None,) * 8
        g0 = g_total.reshape(1)                   # only the total carries gradient
        if g0.dtype != torch.float32 or not g0.is_contiguous():
            g0 = g0.contiguous().float()
        dl = torch.empty_like(lg)
        LIB.call("uh_bce_dice_grad", lg.data_ptr(), mk.data_ptr(), mask_div, None, n, sums.data_ptr(),
                 float(round(n * world)), 1.0, 1.0, g0.data_ptr(), dl.data_ptr(), _stream())
        dl = dl.view(shape)
        if dtype != torch.float32:
            dl = dl.to(dtype)
        return dl, None, None, None, None, None, None, None


class SegLossMulticlassFn(Function):
    """train.py:136-142: CrossEntropyLoss + dice_loss(softmax, one_hot, multiclass=True) on NHWC
    fp32 logits [B,H,W,C] (+ optional w_boundary * boundary_loss on channel 1, train.py:143-147)."""

    @staticmethod
    def forward(ctx, logits_nhwc, mask, w_boundary: float, edge_width: int, edge_weight: float,
                reduce_sums=None, world: int = 1):
        _require_gpu(logits_nhwc, "logits")
        lg = logits_nhwc.float().contiguous()
        B, H, W, C = lg.shape
        npix = B * H * W
        mk = mask.contiguous()
        if mk.dtype != torch.int64:
            mk = mk.long()
        dev = lg.device
        sums = torch.empty(1 + 3 * C, dtype=torch.float32, device=dev)
        ws = loss_workspace(dev)
        LIB.call("uh_ce_dice_sums", lg.data_ptr(), mk.data_ptr(), npix, C, sums.data_ptr(), ws.data_ptr(),
                 ws.numel(), _stream())
        if reduce_sums is not None:
            reduce_sums(sums)
        bl = None
        if w_boundary != 0.0:
            # 4-D path of boundary_loss.py:20-23: channel 1 of the logits, target = mask as float
            bl = boundary_loss_value(lg[..., 1], C, H * W * C, mk.float(), B, H, W, edge_width, edge_weight)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        LIB.call("uh_seg_loss_multiclass_finish", sums.data_ptr(), C, float(round(npix * world)), _p(bl),
                 float(w_boundary), out.data_ptr(), _stream())
        ctx.save_for_backward(lg, mk, sums)
        ctx.meta = (npix, C, world, logits_nhwc.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        lg, mk, sums = ctx.saved_tensors
        npix, C, world, dtype = ctx.meta
        g0 = gout[0:1].contiguous().float()
        dl = torch.empty_like(lg)
        LIB.call("uh_ce_dice_grad", lg.data_ptr(), mk.data_ptr(), npix, C, sums.data_ptr(), float(round(npix * world)),
                 1.0, 1.0, g0.data_ptr(), dl.data_ptr(), _stream())
        if dtype != torch.float32:
            dl = dl.to(dtype)
        return dl, None, None, None, None, None, None


class DiceCoeffFn(Function):
    """dice_coeff (dice_score.py:5-25) over `ngroups` groups of `group_len` elements."""

    @staticmethod
    def forward(ctx, x, t, ngroups: int, group_len: int, eps: float):
        _require_gpu(x, "input")
        xf = x.float().contiguous()
        tf = t.float().contiguous()
        dev = xf.device
        sums = torch.empty(ngroups * 3, dtype=torch.float32, device=dev)
        ws = loss_workspace(dev)
        LIB.call("uh_dice_sums", xf.data_ptr(), tf.data_ptr(), ngroups, group_len, sums.data_ptr(), ws.data_ptr(),
                 ws.numel(), _stream())
        out = torch.empty(1, dtype=torch.float32, device=dev)
        LIB.call("uh_dice_from_sums", sums.data_ptr(), ngroups, float(eps), out.data_ptr(), _stream())
        ctx.save_for_backward(tf, sums)
        ctx.meta = (ngroups, group_len, eps, x.shape, x.dtype)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        tf, sums = ctx.saved_tensors
        ngroups, group_len, eps, shape, dtype = ctx.meta
        s = sums.view(ngroups, 3)
        inter2 = 2 * s[:, 0]
        sets = s[:, 1] + s[:, 2]
        live = sets != 0
        den = sets + eps
        ka = torch.where(live, 2.0 / den, torch.zeros_like(den))
        kb = torch.where(live, -(inter2 + eps) / (den * den), torch.zeros_like(den))
        dx = (ka[:, None] * tf.view(ngroups, group_len) + kb[:, None]) * (g / ngroups)
        return dx.view(shape).to(dtype), None, None, None, None


# ----------------------------------------------------------------------------- inference masks
def argmax_classes(mask_pred: torch.Tensor) -> torch.Tensor:
    """`mask_pred.argmax(dim=1)` of predict.py:27 / evaluate.py:111 for logits [B,C,H,W] -> int64 [B,H,W]."""
    _require_gpu(mask_pred, "logits")
    if mask_pred.dim() != 4:
        raise RuntimeError(f"argmax_classes expects [B,C,H,W] logits, got {tuple(mask_pred.shape)}")
    B, C, H, W = mask_pred.shape
    v = mask_pred.permute(0, 2, 3, 1)
    if v.dtype != torch.float32 or not v.is_contiguous():
        v = v.float().contiguous()
    out = torch.empty(B, H, W, dtype=torch.int64, device=v.device)
    LIB.call("uh_argmax_classes", v.data_ptr(), B * H * W, C, out.data_ptr(), _stream())
    return out


def threshold_mask(logits: torch.Tensor) -> torch.Tensor:
    """`(torch.sigmoid(logits) > 0.5).float()` of evaluate.py:60-62, computed as logits > 0."""
    _require_gpu(logits, "logits")
    v = logits if (logits.dtype == torch.float32 and logits.is_contiguous()) else logits.float().contiguous()
    out = torch.empty(v.shape, dtype=torch.float32, device=v.device)
    LIB.call("uh_threshold_mask", v.data_ptr(), v.numel(), out.data_ptr(), _stream())
    return out
