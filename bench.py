#!/usr/bin/env python3
"""Benchmark of the UNet segmentation train step (BASELINE.json metric: images/sec of one train step,
UNet 1x512x512 -> 1).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = train.py:113-159 of the reference on one resident synthetic batch: forward (bf16 activations,
fp32 accumulate / statistics / master weights), BCE + Dice + 0.25*boundary loss, NaN check, backward,
(RCCL gradient all-reduce when N > 1), clip_grad_norm_(1.0), RMSprop.  N = 1 runs BASELINE config 2
(UNet(1,1,bilinear=True), batch 8); N > 1 keeps 8 images per GPU (weak scaling) unless --global-batch G is given:
then every rank takes G / N images (strong scaling; BASELINE config 3 = --global-batch 32 on 8 GPUs).
Rank 0 prints ONE JSON line.  For N > 1 the weak-scaling line is the headline and a `strong_gb32` object carries the
fixed-global-batch-32 measurement (ms/step, images/s, exposed all-reduce time) of the same ranks.

`python bench.py --gpus N` without a torch.distributed environment starts the N ranks ITSELF (fresh child processes, one
per GPU, rendezvous on 127.0.0.1) before anything in this process has touched the GPU, relays rank 0's JSON line and
exits with the worst rank's code; under `python -m torch.distributed.run ... bench.py --gpus N` (the driver's form) it is
one of the ranks and `--gpus` must equal WORLD_SIZE.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic work (measured by hooks on the reference model): conv FLOPs per 512x512 image
FWD_GFLOP_PER_IMAGE = {True: 319.237, False: 384.735}      # keyed by `bilinear`
TRAIN_GFLOP_PER_IMAGE = {True: 957.41, False: 1153.90}     # fwd + dgrad + wgrad - dgrad(stem)
MFMA_BF16_PEAK_TFLOPS = 2500.0                             # dense, MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3
TRAFFIC_PROFILE = "r05_hbm_traffic_pmc.json"               # HBM bytes per launch from the PMC passes (profiles/README.md)


def source_sha256(name: str) -> str:
    import hashlib
    path = os.path.join(ROOT, "unet-medical-image-contour-segmentation_amd", "csrc", name)
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed GLOBAL batch split evenly over the ranks (strong scaling; BASELINE config 3: 32); 0 = --batch per GPU")
    ap.add_argument("--sync-bn", action="store_true", help="BatchNorm statistics of the global batch (SURVEY 8e option 2)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--fp32", action="store_true", help="fp32 activations (parity path) instead of bf16")
    ap.add_argument("--convt", action="store_true", help="transposed-conv upsample variant (config 5)")
    ap.add_argument("--bf16x3", action="store_true", help="with --fp32: 3x3 conv products on the bf16 matrix pipe (hi/lo splits)")
    ap.add_argument("--cc-loss", action="store_true", help="add connected_component_loss to the loss value (config 5)")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE config 4: 5-level UNet (64..2048/2), 3x1024x1024 in, 4 classes, bilinear (use --batch 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--graph", action="store_true", help="one GPU: replay the whole step from a captured HIP graph (GraphedTrainStepper)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1: skip the global-batch-32 leg (strong_gb32)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained leg")
    ap.add_argument("--no-b4-leg", action="store_true", help="N = 1: skip the 4-images-per-GPU leg (per_gpu_batch4)")
    ap.add_argument("--sustained-seconds", type=float, default=2.5, help="length of the sustained leg")
    ap.add_argument("--no-inference", action="store_true", help="skip the forward-only inference timing (clean train-step profiles)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank flow on a one-GPU box together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--side-stream", action="store_true",
                    help="force backward-weights onto a stream of its own (default: TrainStepper decides per step -- on for bf16 steps "
                         "of >= 2^20 pixels per process: +1.5 %% at batch 8, DESIGN.md)")
    ap.add_argument("--no-side-stream", action="store_true", help="force one stream (A/B runs)")
    return ap.parse_args()


def cpu_baseline(size: int):
    """The reference's train step from stock torch.nn modules (oracle/nn_ref.py: nn.Conv2d / BatchNorm2d / Upsample ... wired as
    unet_parts.py / unet_model.py wire them, stock RMSprop + clip_grad_norm_, pinned to the reference's golden fixtures in fp32
    AND under bf16 autocast: G8 / G15) timed on this host: BASELINE config 1 = batch 2, 1x512x512; fp32 (amp=False, the parity
    leg) and CPU bf16 autocast (= the reference CLI's default --amp on a CUDA-less host, train.py:116,233).  1 warm-up + best of
    3 each.  `functional_restatement` times oracle/step_ref.py (written-out BatchNorm backward, clip and RMSprop:
    the checker of the parity tests) on the same step for comparison."""
    from oracle import nn_ref as N
    from oracle import step_ref as S
    from oracle import unet_ref as U
    # the GPU box gives one GPU a share of 16 host cores (the machine shows 256): oversubscribing slows oneDNN down
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    g = torch.Generator().manual_seed(1)
    images = torch.rand(2, 1, size, size, generator=g).contiguous(memory_format=torch.channels_last)      # train.py:113
    masks = torch.randint(0, 3, (2, size, size), generator=g)
    nsteps = 3
    legs = {}
    for name, amp in (("fp32", False), ("bf16_autocast", True)):
        torch.manual_seed(0)
        model = N.NNUNet(1, 1, True).to(memory_format=torch.channels_last)                               # train.py:262
        stepper = N.NNStepper(model, amp=amp)
        stepper.step(images, masks)                                                                       # warm-up
        best = float("inf")
        for _ in range(nsteps):
            t0 = time.perf_counter()
            stepper.step(images, masks)
            best = min(best, time.perf_counter() - t0)
        legs[name] = {"images_per_sec": round(2.0 / best, 4), "s_per_step": round(best, 3)}
    st = U.init_state(1, 1, True, seed=0)
    st, opt, _ = S.train_step(st, None, images, masks, n_classes=1, bilinear=True)
    t0 = time.perf_counter()
    S.train_step(st, opt, images, masks, n_classes=1, bilinear=True)
    func = time.perf_counter() - t0
    return {"value": legs["fp32"]["images_per_sec"], "unit": "images/sec", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "graph": "torch.nn modules (nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.MaxPool2d, nn.Upsample), channels_last, torch.optim.RMSprop(foreach) "
                     "+ clip_grad_norm_: oracle/nn_ref.py, the reference's own graph restated (the reference's Python does not travel)",
            "sample": f"UNet(1,1,bilinear=True) fp32 (amp=False), batch 2 x 1x{size}x{size}, "
                      f"1 warm-up + best of {nsteps} steps ({legs['fp32']['s_per_step']:.2f} s/step)",
            "bf16_autocast": {"value": legs["bf16_autocast"]["images_per_sec"], "unit": "images/sec",
                              "sample": f"same step under torch.autocast('cpu', bfloat16) (the reference CLI default), "
                                        f"1 warm-up + best of {nsteps} ({legs['bf16_autocast']['s_per_step']:.2f} s/step); "
                                        "pinned by fixture set G15"},
            "functional_restatement": {"value": round(2.0 / func, 4), "unit": "images/sec",
                                       "sample": f"oracle/step_ref.train_step (fp32), one step behind a warm-up ({func:.2f} s/step)"}}


def dice_vs_ref(steps: int = 200, size: int = 64, batch: int = 4, lr: float = 1e-4, seeds=(0, 1, 2, 3, 4), extra=(225, 250)):
    """BASELINE metric's second half, "Dice vs ref", like for like: the same train steps of UNet_T(1,1,bilinear) on seeded
    synthetic ellipse batches run by the CPU oracle (reference restatement, fp32) and by the HIP path (fp32 and bf16) from the
    SAME five initialisations, then the evaluate.py Dice of each on a held-out batch.  Part of the cpu_baseline leg (the oracle
    is the checker here).  RMSprop's sign-like steps (momentum 0.999, batch 4) make the path to the plateau chaotic AND bumpy for
    the reference recipe itself: last-bit differences move the Dice after 200 steps by several points and single trajectories
    dip and recover (profiles/r04_dice_chaos_oracle.txt: the oracle reads 0.98 throughout on three initialisations, 0.95 on one and
    slides 0.92 / 0.87 / 0.92 / 0.86 / 0.86 / 0.74 over steps 150 ... 300 on seed 4).  So every figure is reported per run, the
    headline figures are MEDIANS over the initialisations, and beside the Dice at `steps` each run carries the mean of its Dice
    at `steps` and the `extra` checkpoints (a dip at one checkpoint moves that figure by a third)."""
    import unet_amd
    from oracle import step_ref as S
    from oracle import unet_ref as U
    torch.set_num_threads(min(16, os.cpu_count() or 1))            # tiny tensors: the box's 100+ cores only add overhead
    widths = (8, 16, 32, 64, 128)                                  # UNet_T (unet_model.py:52-82)
    train = [unet_amd.ellipse_batch(batch, size, seed=100 + i) for i in range(4)]
    held = unet_amd.ellipse_batch(8, size, seed=7)
    marks = (steps,) + tuple(extra)
    inits = {seed: U.init_state(1, 1, True, widths=widths, seed=seed) for seed in seeds}

    def med(v):
        return sorted(v)[len(v) // 2]

    out = {"model": f"UNet_T(1,1,bilinear) {batch}x1x{size}x{size}", "steps": steps, "lr": lr, "seeds": list(seeds),
           "avg_checkpoints": list(marks)}
    t0 = time.perf_counter()
    at, avg = [], []
    for seed in seeds:
        st, opt, ds = {k: v.clone() for k, v in inits[seed].items()}, None, []
        for i in range(max(marks)):
            im, mk = train[i % len(train)]
            st, opt, _ = S.train_step(st, opt, im, mk, n_classes=1, bilinear=True, lr=lr)
            if i + 1 in marks:
                ds.append(float(S.evaluate_dice(st, held[0], held[1], n_classes=1, bilinear=True)[0]))
        at.append(round(ds[0], 4))
        avg.append(round(sum(ds) / len(ds), 4))
    out.update({"ref_cpu_fp32": med(at), "ref_cpu_fp32_runs": at, "ref_cpu_fp32_avg": med(avg), "ref_cpu_fp32_avg_runs": avg,
                "ref_cpu_seconds": round(time.perf_counter() - t0, 1)})
    # the reference's OWN bf16 (train.py:116 autocast, its CLI default): the same recipe on the torch.nn graph (oracle/nn_ref.py)
    # under torch.autocast('cpu', bfloat16) -- the like-for-like column for hip_bf16
    from oracle import nn_ref as N
    t0 = time.perf_counter()
    at, avg = [], []
    for seed in seeds:
        twin = N.NNUNet(1, 1, True, widths)
        twin.load_state_dict({k: v.clone() for k, v in inits[seed].items()})
        stepper, ds = N.NNStepper(twin, lr=lr, amp=True), []
        for i in range(max(marks)):
            im, mk = train[i % len(train)]
            stepper.step(im, mk)
            if i + 1 in marks:
                twin.eval()
                with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):                 # evaluate.py:43
                    logits = twin(held[0]).float()
                twin.train()
                pred = (torch.sigmoid(logits.squeeze(1)) > 0.5).float()
                from oracle import losses_ref as L
                ds.append(float(L.dice_coeff(pred, (held[1] // 2).float(), reduce_batch_first=False)))
        at.append(round(ds[0], 4))
        avg.append(round(sum(ds) / len(ds), 4))
    out.update({"ref_cpu_bf16": med(at), "ref_cpu_bf16_runs": at, "ref_cpu_bf16_avg": med(avg), "ref_cpu_bf16_avg_runs": avg,
                "ref_cpu_bf16_seconds": round(time.perf_counter() - t0, 1)})
    dev = torch.device("cuda", torch.cuda.current_device())
    held_set = [{"image": held[0], "mask": held[1]}]
    for name, amp in (("hip_fp32", False), ("hip_bf16", True)):
        at, avg = [], []
        for seed in seeds:
            model = unet_amd.UNet_T(1, 1, bilinear=True)
            model.load_state_dict({k: v.clone() for k, v in inits[seed].items()})
            model = model.to(dev)
            stepper = unet_amd.TrainStepper(model, lr=lr, amp=amp)
            ds = []
            for i in range(max(marks)):
                im, mk = train[i % len(train)]
                stepper.step(im.to(dev), mk.to(dev))
                if i + 1 in marks:
                    ds.append(float(unet_amd.evaluate(model, held_set, dev, amp=amp, postprocess=False)[0]))
            at.append(round(ds[0], 4))
            avg.append(round(sum(ds) / len(ds), 4))
            stepper.optimizer.close()
        out[name] = med(at)
        out[name + "_mean"] = round(sum(at) / len(at), 4)
        out[name + "_runs"] = at
        out[name + "_avg"] = med(avg)
        out[name + "_avg_runs"] = avg
    return out


def baseline_config(args, world: int, B: int, amp: bool, bilinear: bool) -> str:
    """Which line of BASELINE.json's `configs` this run is (0-based index + its text's key words), or 'none' with the reason."""
    gb = B * world
    if args.config4:
        return "configs[3]: 5-level UNet (64->1024), 3x1024x1024 in, 4 classes, bilinear" if args.size == 1024 else "none (config 4 at another size)"
    if args.size != 512:
        return "none (another image size)"
    if not bilinear:
        if not amp and args.cc_loss and gb == 16:
            return f"configs[4]: transposed-conv upsample, fp32, connected_component_loss, batch=16 on {world} GPU(s) (BASELINE: 4)"
        return "none (transposed-conv variant off config 5's recipe: needs --fp32 --cc-loss --global-batch 16)"
    if amp and gb == 32:
        return f"configs[2]: batch=32 data parallel on {world} GPU(s) (BASELINE: 8), gradient all-reduce"
    if amp and world == 1 and B == 8:
        return "configs[1]: UNet bf16 batch=8 on 1 GPU"
    if amp and B == 8:
        return f"configs[1] per GPU on {world} GPUs (weak scaling: 8 images per GPU)"
    if not amp and world == 1 and B == 2:
        return "configs[0] shape (batch=2, fp32) on the GPU (the reference's own leg is the CPU baseline)"
    return "none"


def launch_ranks(n: int) -> int:
    """Start `n` ranks of this script as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
    the contract torch.distributed.run uses) and wait for them.  The parent never initialises the GPU (it only counts
    devices), so no process that holds a GPU context is ever replaced or forked.  Children inherit stdout / stderr: rank 0
    prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    share = "--share-gpu" in sys.argv
    ndev = torch.cuda.device_count()                 # does not create a context
    if not share and ndev < n:
        raise SystemExit(f"bench.py --gpus {n}: this machine shows {ndev} GPU(s) (use --share-gpu --backend gloo only to "
                         "rehearse the multi-rank flow on one GPU)")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
    finally:
        for p in procs:                              # a rank died: do not leave the others waiting at a collective
            if p.poll() is None:
                p.kill()
    return rc


def loader_throughput(dev, B: int, S: int, amp: bool):
    """what it costs to turn DECODED uint8 pixels into the tensors train.py:113-114 hands to the model.
    host = the reference's per-pixel arithmetic (rotation, /255, label remap: oracle/data_prep_ref.py, numpy, one core) +
    the fp32 / int64 host-to-device copies; device = uh_batch_prepare fed from pinned uint8 (copy + one kernel)."""
    import numpy as np
    from oracle import data_prep_ref as R
    from unet_amd.utils.data_loading import prepare_batch_device
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(B, S, S, 1), dtype=np.uint8)
    msk = rng.choice(np.array([0, 128, 255], np.uint8), size=(B, S, S))
    turns = [b % 4 for b in range(B)]
    pin_i, pin_m = torch.from_numpy(img).pin_memory(), torch.from_numpy(msk).pin_memory()
    dt = torch.bfloat16 if amp else torch.float32
    for _ in range(3):
        prepare_batch_device(pin_i, pin_m, turns, device=dev, dtype=dt)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        out = prepare_batch_device(pin_i, pin_m, turns, device=dev, dtype=dt)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    di, dm = pin_i.to(dev), pin_m.to(dev)
    e0.record()
    for _ in range(n):
        prepare_batch_device(di, dm, turns, device=dev, dtype=dt)
    e1.record()
    torch.cuda.synchronize()
    t_kernel = e0.elapsed_time(e1) * 1e-3 / n
    t0 = time.perf_counter()
    hi, hm = R.prepare_batch([im[..., 0] for im in img], msk, turns)
    xi = torch.from_numpy(hi).to(dev).contiguous(memory_format=torch.channels_last)
    xm = torch.from_numpy(hm).to(dev)
    torch.cuda.synchronize()
    t_host = time.perf_counter() - t0
    same = bool(torch.equal(out["image"].float().cpu(), torch.from_numpy(hi).to(dt).float()) and torch.equal(out["mask"].cpu(), torch.from_numpy(hm)))
    return {"what": f"decoded uint8 {B} x {S}x{S} -> model inputs (x4-rotation by index, /255 rule, label remap); PIL decode excluded on both sides",
            "device_images_per_sec": round(B / t_dev, 1), "device_ms_per_batch": round(t_dev * 1e3, 3),
            "device_kernels_only_ms": round(t_kernel * 1e3, 4),
            "host_numpy_images_per_sec": round(B / t_host, 1), "host_ms_per_batch": round(t_host * 1e3, 2),
            "host_cores": 1, "bit_equal": same}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))    # decided before torch.cuda is touched
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the two must agree "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)")
    if not torch.cuda.is_available():
        raise SystemExit(f"rank {rank}/{world}: bench.py needs an MI355X: the train-step path has no CPU fallback")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    import unet_amd
    from unet_amd import ops

    bilinear = not args.convt
    torch.manual_seed(0)
    n_in, n_cls = 1, 1
    if args.config4:
        n_in, n_cls = 3, 4
        if args.size == 512:
            args.size = 1024
        model = unet_amd.UNetDepth(3, 4, True, widths=(64, 128, 256, 512, 1024, 2048))
        TRAIN_GFLOP_PER_IMAGE[True] = 4767.0 * (args.size / 1024.0) ** 2          # 1589 GFLOP fwd per 3x1024^2 image
    else:
        model = unet_amd.UNet(1, 1, bilinear=bilinear)
    model = model.to(memory_format=torch.channels_last).to(dev)
    amp = not args.fp32
    if args.graph:
        if world > 1 or args.cc_loss:
            raise SystemExit("--graph is a single-GPU option without --cc-loss")
        stepper = unet_amd.GraphedTrainStepper(model, lr=1e-5, amp=amp, fp32_mode="bf16x3" if (args.fp32 and args.bf16x3) else "exact",
                                               check_nan=os.environ.get("UH_GRAPH_NO_NAN_CHECK") != "1")
        args.no_kernel_profile = True          # per-launch events cannot be recorded inside a replayed graph
    else:
        stepper = unet_amd.TrainStepper(model, lr=1e-5, amp=amp, wgrad_stream=(False if args.no_side_stream else (True if args.side_stream else None)), cc_loss=args.cc_loss,
                                       fp32_mode="bf16x3" if (args.fp32 and args.bf16x3) else "exact", sync_bn=args.sync_bn)
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} ranks")
        args.batch = args.global_batch // world
    B, S = args.batch, args.size
    sync = stepper.optimizer.sync

    def make_batch(nb: int, seed: int):
        g = torch.Generator().manual_seed(seed)
        im = torch.rand(nb, n_in, S, S, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        mk = torch.randint(0, 3, (nb, S, S), generator=g).to(dev)
        return im, mk

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run_step(im, mk):
        # equal shards: the global batch is known without asking the other ranks (saves SyncBN's blocking host read per step)
        if args.graph:
            return stepper.step(im, mk)
        return stepper.step(im, mk, global_batch=int(im.shape[0]) * world if args.sync_bn else None)

    def measure(im, mk, steps: int, warmup: int):
        """`warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; the time is the
        maximum over the ranks.  -> (seconds, per-step exposed all-reduce ms or None, terms of the last step)"""
        last = None
        if sync is not None:
            sync.time_exposed = False
        for _ in range(warmup):
            last = run_step(im, mk)
        if sync is not None:
            sync.exposed = []
            sync.time_exposed = True
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = run_step(im, mk)
        barrier()
        dt = time.perf_counter() - t0
        exposed = sync.exposed_ms()[-steps:] if sync is not None else None
        if sync is not None:
            sync.time_exposed = False
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, exposed, last

    images, masks = make_batch(B, 1 + rank)
    elapsed, exposed, last = measure(images, masks, args.steps, args.warmup)
    loss = float(last["loss"].detach())

    def exposed_stats(ex):
        return {"exposed_allreduce_ms_per_step": round(sum(ex) / max(len(ex), 1), 4) if ex else None,
                "exposed_allreduce_ms_max": round(max(ex), 4) if ex else None}

    # ---- N > 1: the fixed-global-batch leg (BASELINE config 3 = 32 images over the ranks) beside the weak headline
    strong_gb32 = None
    # (N = 1 runs it too -- 32 images on the one GPU -- so that the 1 -> N scaling of the fixed global batch, the north star's
    # ">= 6x at batch 32 x 512 x 512", is the ratio of two `strong_gb32.images_per_sec` figures measured the same way)
    if not strong and not args.config4 and 32 % world == 0 and not args.no_strong_leg and not args.graph and B == 8 and (world > 1 or S <= 512):
        b2 = 32 // world
        im2, mk2 = make_batch(b2, 101 + rank)
        try:
            dt2, ex2, _ = measure(im2, mk2, args.steps, max(2, args.warmup))
            strong_gb32 = {"global_batch": 32, "per_gpu_batch": b2, "steps": args.steps,
                           "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                           "images_per_sec": round(32 * args.steps / dt2, 2), "scaling": "strong", **exposed_stats(ex2)}
        except RuntimeError as e:
            # a configuration whose tensors pass a kernel's 2 GiB addressing limit at this per-GPU batch (the transposed-conv
            # variant at 32 x 512 x 512 on one GPU: uh_convt2x2_dgrad_mfma refuses it) keeps its headline; every rank fails alike
            torch.cuda.synchronize()
            if sync is not None:
                sync.time_exposed = False
            strong_gb32 = {"global_batch": 32, "per_gpu_batch": b2, "error": str(e)[:200]}
        del im2, mk2

    # ---- dominant-kernel roofline, measured live with events on the launch stream (one extra step)
    def profiled_step(im, mk):
        """One extra step with HIP events around every conv launch on the launch stream.  EVERY rank runs it (it contains the
        gradient / loss-sum collectives); rank 0 records.  -> {family: [calls, seconds, flops]}, [(name, flops, seconds, tag)]"""
        ops.PROFILE.clear()
        ops.PROFILE_ON = rank == 0
        side, stepper.wgrad_stream = stepper.wgrad_stream, None      # time every kernel alone on the launch stream
        # every backward-weights launch is timed WITH its slab reduction, as in rounds 1-3 (the timed steps batch the eighteen
        # reductions into one launch behind the backward pass, which would leave them out of the per-layer figures)
        defer, ops.DEFER_SLABS = ops.DEFER_SLABS, False
        try:
            run_step(im, mk)
            torch.cuda.synchronize()
        finally:
            ops.DEFER_SLABS = defer
        stepper.wgrad_stream = side
        ops.PROFILE_ON = False
        launches = [(n, f, e0.elapsed_time(e1) * 1e-3, t) for n, f, e0, e1, t in ops.PROFILE]
        ops.PROFILE.clear()
        agg = {}
        for name, flops, sec, _tag in launches:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += sec
            a[2] += flops
        return agg, launches

    def down2_in_step(launches, nb):
        """The six conv launches of the 256-channel DoubleConv (down2: 128 -> 256 -> 256 at a quarter of the image extent) where
        they run in the timed workload: one events pair per launch inside the profiled train step, backward-weights including
        its slab reduction, the BatchNorm / pooling kernels of the step between them."""
        q = S // 4
        want = {"fwd_conv1": ("fwd", nb, q, q, 128, 256), "fwd_conv2": ("fwd", nb, q, q, 256, 256),
                "dgrad_conv2": ("dgrad", nb, q, q, 256, 256), "dgrad_conv1": ("dgrad", nb, q, q, 256, 128),
                "wgrad_conv2": ("wgrad", nb, q, q, 256, 256), "wgrad_conv1": ("wgrad", nb, q, q, 128, 256)}
        ins, tot_s, tot_f = {}, 0.0, 0.0
        for key, tag in want.items():
            # up2.3 (256 -> 128 at this extent) is a forward launch WITH statistics: the tag's direction tells it from
            # down2's backward-data; exactly one launch carries each tag
            hits = [(f, sec) for _n, f, sec, t in launches if t == tag]
            if len(hits) != 1:
                raise RuntimeError(f"{key}: {len(hits)} launches tagged {tag}")
            f, sec = hits[0]
            ins[key] = {"ms": round(sec * 1e3, 4), "tflops": round(f / sec / 1e12, 1)}
            tot_s += sec
            tot_f += f
        ins["all_six"] = {"ms": round(tot_s * 1e3, 4), "tflops": round(tot_f / tot_s / 1e12, 1),
                          "frac_of_peak": round(tot_f / tot_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
        ins["what"] = f"one launch each, inside the profiled train step at {nb} images per GPU (single events pair per launch)"
        return ins

    in_step_ok = amp and bilinear and not args.config4 and n_in == 1
    roof = None
    kernels = None
    agg = launches = None
    if not args.no_kernel_profile:
        agg, launches = profiled_step(images, masks)
    if rank == 0 and not args.no_kernel_profile:
        kernels = {k: {"calls": v[0], "ms": round(v[1] * 1e3, 3), "tflops": round(v[2] / v[1] / 1e12, 1) if v[1] > 0 else 0.0}
                   for k, v in agg.items()}
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        peak = MFMA_BF16_PEAK_TFLOPS if amp else MFMA_F32_PEAK_TFLOPS
        ach = dom[1][2] / dom[1][1] / 1e12
        # HBM bytes per launch of this kernel family from the PMC passes committed under profiles/ (rocprofv3 cannot be
        # driven from inside the process); null for configurations that were not profiled
        traffic = None
        traffic_note = "not profiled for this configuration"
        try:
            if amp and bilinear and B == 8 and S == 512 and not args.config4:
                tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)))
                # the profile is stamped with the sha256 of the kernel source it was measured on: a stale profile (the
                # kernels changed since) yields null instead of a number that no longer describes the binary
                if tj.get("source_sha256", {}).get("conv3x3.hip") == source_sha256("conv3x3.hip"):
                    traffic = tj.get(dom[0] + "_per_launch", {}).get("hbm_MB")
                    traffic = None if traffic is None else round(traffic * 1e6)
                    traffic_note = f"profiles/{TRAFFIC_PROFILE} (separate --pmc FETCH_SIZE / WRITE_SIZE passes)"
                else:
                    traffic_note = f"profiles/{TRAFFIC_PROFILE} is stale: conv3x3.hip changed since it was measured"
        except Exception as e:
            traffic, traffic_note = None, f"no usable traffic profile ({e.__class__.__name__})"
        roof = {"kernel": dom[0], "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_note,
                "avg_launch_ms": round(dom[1][1] / dom[1][0] * 1e3, 4), "launches_per_step": dom[0] and dom[1][0],
                "timing": "HIP events around every launch in ONE extra step run on the launch stream alone (every kernel by itself, "
                          "whatever config.streams says about the timed steps)"}
        if args.fp32 and args.bf16x3 and dom[0].startswith("conv3x3_"):
            # fp32 tensors, products on the bf16 matrix pipe as three bf16 products each (hi*hi + hi*lo + lo*hi): the pipe that
            # bounds the kernel is the bf16 one and it executes 3x the algorithmic FLOPs -- quoted against THAT peak (dividing
            # the algorithmic rate by the fp32 MFMA peak gave a "fraction" of 1.6: not a measurement)
            roof.update({"peak": MFMA_BF16_PEAK_TFLOPS, "pipe_tflops": round(3 * ach, 1),
                         "frac": round(3 * ach / MFMA_BF16_PEAK_TFLOPS, 4),
                         "note": "achieved = algorithmic fp32 FLOP/s; frac = 3 x achieved / bf16 MFMA peak (bf16x3 split products)"})
        # every launch of the forward / backward-data kernel, whatever its epilogue carries: the eight backward-data launches
        # with the fused BatchNorm-backward sums are a family of their own above (their time includes a read of the BatchNorm
        # input and the sums), so the plain family alone would flatter the kernel
        fam = [v for k, v in agg.items() if k in ("conv3x3_fwd_mfma", "conv3x3_dgrad_bnsum_mfma")]
        if dom[0] == "conv3x3_fwd_mfma" and fam:
            f_all, s_all, n_all = sum(v[2] for v in fam), sum(v[1] for v in fam), sum(v[0] for v in fam)
            roof["all_launches"] = {"launches_per_step": n_all, "ms": round(s_all * 1e3, 3),
                                    "achieved": round(f_all / s_all / 1e12, 1)}
            roof["all_launches_frac"] = round(f_all / s_all / 1e12 / peak, 4)
        # the layer the north-star names: the 256-channel DoubleConv (down2: 128->256->256 at 128x128, batch 8)
        try:
            kernels["double_conv_256"] = ops.bench_double_conv(B, S // 4, S // 4, 128, 256, torch.bfloat16 if amp else torch.float32)
        except Exception as e:                      # an optional extra must never cost the headline line
            kernels["double_conv_256"] = {"error": repr(e)}
        # ... and the same six launches where they run in the timed workload: down2 of the profiled train step above
        if in_step_ok:
            try:
                kernels["double_conv_256_in_step"] = down2_in_step(launches, B)
            except Exception as e:
                kernels["double_conv_256_in_step"] = {"error": repr(e)}
        # the same six kernels at the extent the layer has in BASELINE config 3's global batch (32 images): eight tiles per
        # resident workgroup instead of two, the 75 MB of backward-weights slabs amortised over 4x the pixels
        if amp and B == 8 and S == 512 and not args.config4:
            try:
                kernels["double_conv_256_batch32"] = ops.bench_double_conv(32, S // 4, S // 4, 128, 256, torch.bfloat16, iters=10)
            except Exception as e:
                kernels["double_conv_256_batch32"] = {"error": repr(e)}

    # ---- N > 1: the kernel figure at the strong leg's per-GPU batch (config 3 runs 4 images per GPU on 8 GPUs)
    if strong_gb32 is not None and "error" not in strong_gb32 and not args.no_kernel_profile and in_step_ok:
        b2 = strong_gb32["per_gpu_batch"]
        im2, mk2 = make_batch(b2, 101 + rank)
        _agg2, launches2 = profiled_step(im2, mk2)
        if rank == 0:
            try:
                strong_gb32["double_conv_256_in_step"] = down2_in_step(launches2, b2)
            except Exception as e:
                strong_gb32["double_conv_256_in_step"] = {"error": repr(e)}
        del im2, mk2

    # ---- N = 1: config 3's per-GPU workload (4 images per GPU) on this GPU, so that the fixed per-step costs that decide the
    # 1 -> 8 scaling of a fixed global batch are in the one-GPU record too
    per_gpu_batch4 = None
    if world == 1 and not strong and B == 8 and not args.config4 and not args.no_b4_leg:
        im4, mk4 = make_batch(4, 201)
        dt4, _, _ = measure(im4, mk4, args.steps, max(2, args.warmup))
        per_gpu_batch4 = {"per_gpu_batch": 4, "steps": args.steps, "ms_per_step": round(dt4 / args.steps * 1e3, 3),
                          "images_per_sec": round(4 * args.steps / dt4, 2),
                          "vs_batch8_images_per_sec": round((4 * args.steps / dt4) / (B * args.steps / elapsed), 4)}
        if not args.no_kernel_profile and in_step_ok:
            _agg4, launches4 = profiled_step(im4, mk4)
            try:
                per_gpu_batch4["double_conv_256_in_step"] = down2_in_step(launches4, 4)
            except Exception as e:
                per_gpu_batch4["double_conv_256_in_step"] = {"error": repr(e)}
        del im4, mk4

    # ---- sustained leg: a training job is minutes of back-to-back steps, the headline above is a 0.2 s window.  >= 2 s of
    # continuous steps (every rank takes part: the steps hold collectives), and the 256-channel DoubleConv timed again
    # right behind them, so that the spread the chip's power state causes is in the record
    sustained = None
    if not args.no_sustained:
        n_sus = max(args.steps, int(args.sustained_seconds / max(elapsed / args.steps, 1e-4)) + 1)
        dt_s, ex_s, _ = measure(images, masks, n_sus, 0)
        sustained = {"steps": n_sus, "seconds": round(dt_s, 3), "ms_per_step": round(dt_s / n_sus * 1e3, 3),
                     "images_per_sec": round(world * B * n_sus / dt_s, 2)}
        if ex_s:
            sustained.update(exposed_stats(ex_s))
        if rank == 0 and kernels is not None:
            try:
                kernels["double_conv_256_after_sustained"] = ops.bench_double_conv(
                    B, S // 4, S // 4, 128, 256, torch.bfloat16 if amp else torch.float32)
            except Exception as e:
                kernels["double_conv_256_after_sustained"] = {"error": repr(e)}

    if rank == 0:
        ips = world * B * args.steps / elapsed
        out = {
            "metric": f"images/sec (train step) UNet {n_in}x{S}x{S}->{n_cls}",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "bf16" if amp else ("f32 (3x3 fwd/dgrad as bf16x3 split products)" if args.bf16x3 else "f32"),
            "data": "synthetic",
            "config": {"workload": (f"UNet({n_in},{n_cls},bilinear={bilinear}{', depth 5' if args.config4 else ''}) train step "
                                    f"({'BCE' if n_cls == 1 else 'CE'}+Dice+boundary, clip 1.0, RMSprop), "
                                    f"{B} x {n_in}x{S}x{S} per GPU, global batch {B * world}"),
                       "baseline_config": baseline_config(args, world, B, amp, bilinear),
                       "parallelism": f"dp{world}", "global_batch": B * world, "per_gpu_batch": B,
                       "bn": "global-batch statistics (SyncBN)" if args.sync_bn else "per-rank batch statistics",
                       "dice": "global-batch sums (all-reduced)",
                       "streams": ("backward-weights on a side stream" if stepper._side_for(images) is not None else "one stream")},
            "loss": round(loss, 6),
            "train_tflops_per_gpu": round(ips / world * TRAIN_GFLOP_PER_IMAGE[bilinear] / 1e3, 1),
            "conv_roofline_frac_step": round(ips / world * TRAIN_GFLOP_PER_IMAGE[bilinear] / 1e3 /
                                             (MFMA_BF16_PEAK_TFLOPS if amp else MFMA_F32_PEAK_TFLOPS), 4),
            "roofline": roof, "kernels": kernels, "sustained": sustained,
        }
        if strong_gb32 is not None:
            out["strong_gb32"] = strong_gb32
        if per_gpu_batch4 is not None:
            out["per_gpu_batch4"] = per_gpu_batch4
        if world > 1:
            import torch.distributed as dist
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:
                rccl = repr(e)
            out["collective"] = {
                "backend": dist.get_backend(), "world": dist.get_world_size(), "rccl_version": rccl,
                "per_gpu_batch": B, "global_batch": B * world,
                "grad_bytes": int(stepper.optimizer.flat_g.numel() * 4), "grad_buckets": len(sync.buckets),
                "bucket_bytes": [int((e - s_) * 4) for s_, e in sync.buckets],
                "allreduce": "SUM of the flat fp32 gradient, one async all_reduce per bucket in backward-ready order from a stream of its own",
                "sync_bn": (None if not args.sync_bn else
                            ("own communicator (UH_SYNCBN_OWN_GROUP=1)" if stepper.bn_group is not None else "gradient communicator")),
                # time the launch stream spent waiting for the gradient all-reduce in front of clip + RMSprop (two events
                # around the waits, nothing else between them): what the overlap with the encoder backward did NOT hide
                **exposed_stats(exposed),
            }
        if world == 1 and not args.no_inference:
            # forward-only inference (model.eval(): running statistics folded into the conv epilogue)
            try:
                model.eval()
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    for _ in range(3):
                        model(images)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        model(images)
                    e1.record()
                    torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                out["inference"] = {"images_per_sec": round(B / ms * 1e3, 1), "ms_per_batch": round(ms, 3), "batch": B,
                                    "what": "eval-mode forward, fused conv+BN(running stats)+ReLU kernels, logits only"}
            except Exception as e:                  # optional extras never cost the headline line
                out["inference"] = {"error": repr(e)}
            model.train()
        if world == 1 and not args.no_cpu_baseline and n_in == 1:
            try:
                out["loader"] = loader_throughput(dev, B, S, amp)
            except Exception as e:
                out["loader"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(S)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
            try:
                out["cpu_baseline"]["dice_vs_ref"] = dice_vs_ref()
            except Exception as e:
                out["cpu_baseline"]["dice_vs_ref"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
