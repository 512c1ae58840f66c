"""Data-parallel step on the GPU path, world_size 2.  RCCL refuses two ranks on one device, so on the 1-GPU test box the
two ranks share cuda:0 over the gloo backend: this exercises exactly the code that runs under RCCL (flat gradient
buckets launched from the backward callbacks/hooks, side-stream wgrad, all-reduced loss sums, fused optimizer) with
another transport.  Invariants: identical parameters on both ranks after each step, global-batch Dice/BCE values
identical on both ranks, and the synchronised gradient = mean of the two ranks' local gradients."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import unet_amd
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = unet_amd.UNet_S(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
        # the opt-in side stream for backward-weights: its events gate the per-bucket all-reduce (dp.BucketedGradSync)
        stepper = unet_amd.TrainStepper(model, lr=1e-4, amp=True, wgrad_stream=True)
        assert stepper.world == 2 and stepper.optimizer.sync is not None and len(stepper.optimizer.sync.buckets) >= 1
        g = torch.Generator().manual_seed(10 + rank)
        images = torch.rand(2, 1, 96, 96, generator=g).to(dev)
        masks = torch.randint(0, 3, (2, 96, 96), generator=g).to(dev)
        # local gradient of this rank (no sync) for the mean check: same loss definition, world-aware sums
        for step in range(2):
            terms = stepper.step(images, masks)
            torch.cuda.synchronize()
            flat = stepper.optimizer.flat_p.detach().cpu()
            gathered = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(gathered, flat)
            assert torch.equal(gathered[0], gathered[1]), f"parameters diverged at step {step}"
            vals = torch.tensor([float(terms["dice"]), float(terms["bce"]), float(terms["loss"] - 0.25 * terms["boundary"])])
            allv = [torch.zeros_like(vals) for _ in range(world)]
            dist.all_gather(allv, vals)
            assert torch.allclose(allv[0], allv[1], rtol=1e-6), (allv, "global-batch loss terms differ between ranks")
            assert torch.isfinite(vals).all()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_ranks_share_one_gpu_over_gloo():
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]   # (below the GPU box's 420 s silence limit: a hung rank fails the test instead of the run)
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _syncbn_worker(rank, world, port, q, cuts=(0, 2, 4)):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import unet_amd
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
        stepper = unet_amd.TrainStepper(model, lr=1e-4, amp=False, sync_bn=True)
        im, mk = unet_amd.ellipse_batch(4, 64, seed=21)
        lo, hi = cuts[rank], cuts[rank + 1]
        out = None
        for _ in range(2):
            out = stepper.step(im[lo:hi].to(dev), mk[lo:hi].to(dev))
        torch.cuda.synchronize()
        # numpy (pickled by value): torch tensors travel as shared-memory handles that die with this process
        q.put((rank, "ok", stepper.optimizer.flat_p.cpu().numpy(), stepper.optimizer.flat_g.cpu().numpy(), float(out["bce"]),
               float(out["dice"]), {k: v.cpu().numpy() for k, v in model.state_dict().items() if "running" in k}))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("cuts", [(0, 2, 4), (0, 3, 4)], ids=["equal-shards", "ragged-3+1"])
def test_sync_bn_data_parallel_equals_single_process(cuts):
    """TrainStepper(sync_bn=True): two ranks with half the batch each -- or with 3 and 1 images (the last batch of an
    epoch) -- == one process with the whole batch: parameters, (clipped) gradients, BatchNorm running statistics, BCE and
    Dice values.  This is the exact global-batch parity option of SURVEY.md 8(e); the default per-rank BatchNorm is what
    stock DDP does."""
    import unet_amd
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    stepper = unet_amd.TrainStepper(model, lr=1e-4, amp=False)
    p0 = stepper.optimizer.flat_p.cpu().clone()
    im, mk = unet_amd.ellipse_batch(4, 64, seed=21)
    ref = None
    for _ in range(2):
        ref = stepper.step(im.to(dev), mk.to(dev))
    torch.cuda.synchronize()
    ref_p, ref_g = stepper.optimizer.flat_p.cpu(), stepper.optimizer.flat_g.cpu()
    ref_run = {k: v.cpu() for k, v in model.state_dict().items() if "running" in k}
    ref_bce, ref_dice = float(ref["bce"]), float(ref["dice"])
    del stepper, model
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q, cuts)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]   # (below the GPU box's 420 s silence limit: a hung rank fails the test instead of the run)
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    for r in res:
        _, _, fp, fg, bce, dice, run = r
        fp, fg = torch.from_numpy(fp), torch.from_numpy(fg)
        run = {k: torch.from_numpy(v) for k, v in run.items()}
        assert abs(bce - ref_bce) < 1e-5 * max(1.0, abs(ref_bce)) and abs(dice - ref_dice) < 1e-5
        gerr = float((fg - ref_g).norm() / ref_g.norm())
        # RMSprop divides by sqrt(mean g^2): elements whose gradient is ~0 take sign-like steps of size ~lr that fp32
        # round-off can flip, so the UPDATE is compared in L2 and every element to within one step size
        uerr = float(((fp - p0) - (ref_p - p0)).norm() / (ref_p - p0).norm())
        perr = float((fp - ref_p).abs().max())
        assert gerr < 1e-3, f"gradient rel L2 {gerr:.3e}"
        # (one RMSprop step at lr 1e-4 moves an element by up to lr / sqrt(1 - alpha) = 1e-3)
        assert uerr < 2e-2 and perr < 5e-4, f"update rel L2 {uerr:.3e}, max abs diff {perr:.3e}"
        for k, v in ref_run.items():
            if k.endswith("running_mean"):      # a mean is known to a fraction of the data's spread, not of its own size
                std = ref_run[k.replace("running_mean", "running_var")].sqrt()
                err = float(((run[k] - v).abs() / (1e-4 * std + 1e-6)).max())
            else:
                err = float(((run[k] - v).abs() / (1e-4 * v.abs() + 1e-6)).max())
            assert err <= 1.0, f"{k}: {err:.2f} x tolerance"
    assert (res[0][2] == res[1][2]).all()


# ------------------------------------------------------------------------------------------ the untested DP pieces
def _spawn(target, *extra, world=2, timeout=300):       # (below the GPU box's 420 s silence limit)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=timeout) for _ in procs]
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:               # a rank that hangs must not outlive the test (it holds the GPU)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    assert all(r[1] == "ok" for r in res), res
    return sorted(res)


def _nan_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import unet_amd
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
        stepper = unet_amd.TrainStepper(model, lr=1e-4, amp=False)
        im, mk = unet_amd.ellipse_batch(4, 64, seed=31)
        im, mk = im[rank * 2:rank * 2 + 2].to(dev), mk[rank * 2:rank * 2 + 2].to(dev)
        stepper.step(im, mk)
        before = stepper.optimizer.flat_p.detach().clone()
        bad = im.clone()
        if rank == 1:
            bad[0, 0, 5, 7] = float("nan")            # ONE rank sees the NaN: every rank must take the same decision
        raised = False
        try:
            stepper.step(bad, mk)
        except RuntimeError as e:
            raised = "NaN loss" in str(e)
        torch.cuda.synchronize()
        untouched = bool(torch.equal(before, stepper.optimizer.flat_p))
        t = stepper.step(im, mk)                      # the next step is clean: no stale collective, no stale gradient
        torch.cuda.synchronize()
        flat = stepper.optimizer.flat_p.detach().cpu()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = bool(torch.equal(gathered[0], gathered[1]))
        finite = bool(torch.isfinite(flat).all()) and bool(torch.isfinite(t["loss"]))
        moved = not bool(torch.equal(before.cpu(), flat))
        dist.destroy_process_group()
        q.put((rank, "ok", raised, untouched, same, finite, moved))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_nan_on_one_rank_aborts_every_rank_and_the_next_step_is_clean():
    for _, _, raised, untouched, same, finite, moved in _spawn(_nan_worker):
        assert raised, "a rank did not raise RuntimeError('Fatal: NaN loss detected!')"
        assert untouched, "the aborted step changed parameters"
        assert same and finite and moved


def _eval_worker(rank, world, port, q, postprocess):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import unet_amd
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
        batches = [dict(zip(("image", "mask"), unet_amd.ellipse_batch(2, 64, seed=40 + i))) for i in range(4)]
        mine = batches[rank::world]                       # every rank evaluates its shard of the validation set
        d, dp_, mn = unet_amd.evaluate(model, mine, dev, amp=False, postprocess=postprocess)
        dist.destroy_process_group()
        q.put((rank, "ok", float(d), float(dp_), float(mn)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("postprocess", [False, True])
def test_sharded_evaluate_returns_the_metric_of_the_whole_set(postprocess):
    import unet_amd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    batches = [dict(zip(("image", "mask"), unet_amd.ellipse_batch(2, 64, seed=40 + i))) for i in range(4)]
    d, dp_, mn = (float(v) for v in unet_amd.evaluate(model, batches, dev, amp=False, postprocess=postprocess))
    for _, _, rd, rdp, rmn in _spawn(_eval_worker, postprocess):
        assert abs(rd - d) < 1e-6 and abs(rdp - dp_) < 1e-6 and abs(rmn - mn) < 1e-6, ((rd, rdp, rmn), (d, dp_, mn))


def test_parameter_without_gradient_is_skipped_like_torch_optim():
    """A parameter that receives no gradient (frozen after construction / unused in this step) must stay exactly where it
    is -- torch.optim skips parameters whose .grad is None; weight decay / momentum must not move it -- while every other
    parameter takes the same step as in an unfrozen run whose frozen-parameter gradient is removed from the norm."""
    import unet_amd
    dev = torch.device("cuda:0")
    im, mk = unet_amd.ellipse_batch(2, 64, seed=5)
    torch.manual_seed(0)
    model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    opt = unet_amd.FusedRMSprop(model.parameters(), lr=1e-3, weight_decay=1e-2)     # large decay: drift would be visible
    frozen = [model.up2.conv.double_conv[4].weight, model.up2.conv.double_conv[4].bias, model.down1.maxpool_conv[1].double_conv[0].weight]
    for step in range(3):
        for p in frozen:
            p.requires_grad_(step == 0)          # step 0 builds optimizer state for them, then they are frozen
        snap = [p.detach().clone() for p in frozen]
        others = [p.detach().clone() for p in model.parameters() if all(p is not f for f in frozen)]
        model.train()
        opt.zero_grad()
        unet_amd.seg_loss(model(im.to(dev)), mk.to(dev), 1)["loss"].backward()
        opt.step()
        torch.cuda.synchronize()
        if step > 0:
            for p, s0 in zip(frozen, snap):
                assert torch.equal(p.detach(), s0), "a parameter without a gradient moved"
        now = [p.detach() for p in model.parameters() if all(p is not f for f in frozen)]
        assert any(not torch.equal(a, b) for a, b in zip(now, others))
    # against stock torch.optim.RMSprop + clip_grad_norm_ on a twin model driven the same way
    torch.manual_seed(0)
    twin = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    topt = torch.optim.RMSprop(twin.parameters(), lr=1e-3, weight_decay=1e-2, momentum=0.999)
    tfrozen = [twin.up2.conv.double_conv[4].weight, twin.up2.conv.double_conv[4].bias, twin.down1.maxpool_conv[1].double_conv[0].weight]
    for step in range(3):
        for p in tfrozen:
            p.requires_grad_(step == 0)
        twin.train()
        topt.zero_grad(set_to_none=True)
        unet_amd.seg_loss(twin(im.to(dev)), mk.to(dev), 1)["loss"].backward()
        torch.nn.utils.clip_grad_norm_([p for p in twin.parameters() if p.grad is not None], 1.0)
        topt.step()
    for (k, a), (_, b) in zip(model.named_parameters(), twin.named_parameters()):
        err = float((a.detach() - b.detach()).abs().max())
        assert err <= 2e-3, f"{k}: {err:.2e} from stock RMSprop after three steps (one step moves an element by up to 1e-2)"


def test_wgrad_stream_choice_is_per_stepper():
    import unet_amd
    from unet_amd import ops
    dev = torch.device("cuda:0")
    im, mk = unet_amd.ellipse_batch(2, 64, seed=6)
    a = unet_amd.TrainStepper(unet_amd.UNet_T(1, 1, bilinear=True).to(dev), amp=False, wgrad_stream=True)
    b = unet_amd.TrainStepper(unet_amd.UNet_T(1, 1, bilinear=True).to(dev), amp=False, wgrad_stream=False)
    a.step(im.to(dev), mk.to(dev))
    assert ops.WGRAD_STREAM is a.wgrad_stream and a.wgrad_stream is not None
    b.step(im.to(dev), mk.to(dev))
    assert ops.WGRAD_STREAM is None
    a.step(im.to(dev), mk.to(dev))
    assert ops.WGRAD_STREAM is a.wgrad_stream           # constructing / stepping b did not take a's side stream away
    torch.cuda.synchronize()


def test_side_stream_is_chosen_per_step_by_default():
    """TrainStepper() without a wgrad_stream argument: backward-weights goes to the side stream for bf16 steps of at least 2^20 pixels
    per process (where it measured +1.5 % at batch 8 and +2.2 % at batch 4) and stays on the launch stream for small or fp32
    steps (exact fp32 lost 3 %, batch 2 is host-bound); the step is the same step either way."""
    import unet_amd
    from unet_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    st = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
    assert st.wgrad_stream is not None and st._side_auto
    small = torch.zeros(2, 1, 64, 64, device=dev)
    big = torch.zeros(4, 1, 512, 512, device=dev)
    assert st._side_for(small) is None and st._side_for(big) is st.wgrad_stream
    fp = unet_amd.TrainStepper(unet_amd.UNet_T(1, 1, bilinear=True).to(dev), amp=False)
    assert fp._side_for(big) is None
    assert unet_amd.TrainStepper(unet_amd.UNet_T(1, 1, bilinear=True).to(dev), amp=True, wgrad_stream=False)._side_for(big) is None
    forced = unet_amd.TrainStepper(unet_amd.UNet_T(1, 1, bilinear=True).to(dev), amp=True, wgrad_stream=True)
    assert forced._side_for(small) is forced.wgrad_stream
    # a step on each side of the threshold leaves the matching stream installed, and both give finite losses
    im, mk = unet_amd.ellipse_batch(2, 64, seed=6)
    t = st.step(im.to(dev), mk.to(dev))
    assert ops.WGRAD_STREAM is None and math.isfinite(float(t["loss"]))
    im, mk = unet_amd.ellipse_batch(4, 512, seed=7)
    t = st.step(im.to(dev), mk.to(dev))
    assert ops.WGRAD_STREAM is st.wgrad_stream and math.isfinite(float(t["loss"]))
    torch.cuda.synchronize()


def _stale_pattern_worker(rank, world, port, q, mode):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import unet_amd
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
        opt = unet_amd.FusedRMSprop(model.parameters(), lr=1e-4)
        assert opt.sync is not None
        reduce_sums = unet_amd.dp.make_sum_reducer(None)
        frozen = model.up2.conv.double_conv[4].weight            # a BatchNorm gamma: its gradient goes through the accumulate hook
        im, mk = unet_amd.ellipse_batch(4, 64, seed=5)
        im, mk = im[rank * 2:rank * 2 + 2].to(dev), mk[rank * 2:rank * 2 + 2].to(dev)
        err = None
        for step in range(3):
            # "rank": frozen on rank 1 only, from the first step; "later": on BOTH ranks, from step 1 (legal: torch.optim + DDP allow
            # it); "later-one-rank": on rank 1 only, from step 1
            off = (mode == "rank" and rank == 1) or (mode == "later" and step >= 1) or (mode == "later-one-rank" and rank == 1 and step >= 1)
            frozen.requires_grad_(not off)
            model.train()
            try:
                unet_amd.train_step(model, opt, im, mk, amp=False, reduce_sums=reduce_sums, world=world)
            except RuntimeError as e:
                err = (step, str(e))
                break
        torch.cuda.synchronize()
        q.put((rank, "ok", err))
        q.close()
        q.join_thread()      # (the queue's feeder thread must have flushed the item before the process goes away)
        os._exit(0)          # (skip the interpreter's teardown: nothing here may wait for a collective a peer will never join)
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("mode", ["rank", "later", "later-one-rank"])
def test_a_parameter_stale_on_one_rank_only_is_refused(mode):
    """torch.optim behind DDP updates a parameter on every rank or on none (the all-reduce makes .grad non-None everywhere); the
    fused optimizer skips locally stale slices, so a parameter that is fresh on one rank and stale on another would diverge
    silently.  The pattern is agreed on the first step (both ranks raise before any update).  Later it may change on every rank
    alike (freezing a layer mid-run); a change on one rank only is caught by the checksum that rides in the next step's NaN-flag
    all-reduce -- a collective every rank issues every step -- and BOTH ranks raise together, one step late (no rank may
    be left waiting at a collective its peer has abandoned)."""
    res = _spawn(_stale_pattern_worker, mode)
    for rank, status, err in res:
        assert status == "ok", (rank, status)
        if mode == "rank":
            assert err is not None and err[0] == 0 and "some ranks" in err[1], (rank, err)
        elif mode == "later":
            assert err is None, (rank, err)
        else:
            assert err is not None and err[0] == 2 and "previous step" in err[1], (rank, err)


# ------------------------------------------------------------------------------------------ the real backend, one rank
def _rccl_one_rank_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["UH_DP_FORCE_SYNC"] = "1"          # run the collectives although the group has one rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        import unet_amd
        torch.manual_seed(0)
        model = unet_amd.UNet_S(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
        stepper = unet_amd.TrainStepper(model, lr=1e-4, amp=True, wgrad_stream=True)
        sync = stepper.optimizer.sync
        assert sync is not None and sync.active and dist.get_backend() == "nccl"
        sync.time_exposed = True
        im, mk = unet_amd.ellipse_batch(4, 96, seed=3)
        for _ in range(3):
            t = stepper.step(im.to(dev), mk.to(dev))
        torch.cuda.synchronize()
        ex = sync.exposed_ms()
        q.put((rank, "ok", stepper.optimizer.flat_p.cpu().numpy(), float(t["loss"]), len(sync.buckets), len(ex)))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_bucketed_gradient_sync_runs_over_rccl_with_one_rank():
    """RCCL refuses two ranks on one device, so on a one-GPU box the N > 1 tests above travel over gloo.  This one runs the
    SAME code against the real backend with a one-rank group (UH_DP_FORCE_SYNC=1): loss-sum all-reduces, per-bucket asynchronous
    all-reduces issued from the side stream's events, handle waits in front of the fused optimizer -- every collective is the
    identity, so three steps must leave the parameters bit-identical to a plain single-process run."""
    import unet_amd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unet_amd.UNet_S(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
    st = unet_amd.TrainStepper(model, lr=1e-4, amp=True, wgrad_stream=True)
    im, mk = unet_amd.ellipse_batch(4, 96, seed=3)
    for _ in range(3):
        t = st.step(im.to(dev), mk.to(dev))
    torch.cuda.synchronize()
    ref_p, ref_loss = st.optimizer.flat_p.cpu(), float(t["loss"])
    st.optimizer.close()
    (_, _, p, loss, nbuckets, nex), = _spawn(_rccl_one_rank_worker, world=1)
    assert nbuckets >= 1 and nex == 3
    assert loss == ref_loss and torch.equal(torch.from_numpy(p), ref_p)


def _rccl_one_rank_syncbn_worker(rank, world, port, q, own_group):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["UH_DP_FORCE_SYNC"] = "1"
        os.environ["UH_SYNCBN_OWN_GROUP"] = "1" if own_group else "0"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        import unet_amd
        torch.manual_seed(0)
        model = unet_amd.UNet_S(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
        stepper = unet_amd.TrainStepper(model, lr=1e-4, amp=False, wgrad_stream=True, sync_bn=True)
        assert stepper.sync_bn is not None and (stepper.bn_group is not None) == bool(own_group)
        im, mk = unet_amd.ellipse_batch(4, 96, seed=3)
        for _ in range(3):
            t = stepper.step(im.to(dev), mk.to(dev))           # no global_batch: the batch-size all-reduce runs too
        torch.cuda.synchronize()
        out = (rank, "ok", stepper.optimizer.flat_p.cpu().numpy(), float(t["loss"].detach()))
        stepper.close()                                        # destroys the BatchNorm communicator when there is one
        assert stepper.bn_group is None
        dist.barrier()                                         # the gradient communicator still works afterwards
        q.put(out)
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("own_group", [False, True])
def test_sync_bn_collectives_run_over_rccl_with_one_rank(own_group):
    """SyncBN's all_gather / all_reduce per BatchNorm layer against the real backend (one-rank group, UH_DP_FORCE_SYNC=1), on the
    gradient communicator (default) and on a communicator of their own (UH_SYNCBN_OWN_GROUP=1: two communicators issuing
    collectives from one process, the bucket all-reduces from the side stream's events).  Every collective is the identity; the
    statistics take the merge path (finalize -> gather -> finalize), so the parameters agree with the plain run to round-off."""
    import unet_amd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unet_amd.UNet_S(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
    st = unet_amd.TrainStepper(model, lr=1e-4, amp=False, wgrad_stream=True)
    im, mk = unet_amd.ellipse_batch(4, 96, seed=3)
    for _ in range(3):
        t = st.step(im.to(dev), mk.to(dev))
    torch.cuda.synchronize()
    ref_p, ref_loss = st.optimizer.flat_p.cpu(), float(t["loss"])
    st.close()
    (_, _, p, loss), = _spawn(_rccl_one_rank_syncbn_worker, own_group, world=1)
    # (RMSprop's first steps are sign-like: lr * g / sqrt(0.01 g^2) -- a last-bit difference in a near-zero gradient moves that
    # parameter by 2 * 10 * lr, so three steps agree to ~1e-3 of the parameter norm, not to round-off)
    assert abs(loss - ref_loss) <= 2e-3 * abs(ref_loss), (loss, ref_loss)
    d = (torch.from_numpy(p) - ref_p).norm() / ref_p.norm()
    assert d < 2e-3, d


# ------------------------------------------------------------------------------------------ full-width bf16 under SyncBN
def _fullwidth_syncbn_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import unet_amd
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        model = unet_amd.UNet(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
        stepper = unet_amd.TrainStepper(model, lr=1e-5, amp=True, sync_bn=True)
        im, mk = unet_amd.ellipse_batch(4, 64, seed=77)
        lo, hi = rank * 2, rank * 2 + 2
        t = None
        for _ in range(2):
            t = stepper.step(im[lo:hi].to(dev), mk[lo:hi].to(dev), global_batch=4)
        torch.cuda.synchronize()
        q.put((rank, "ok", stepper.optimizer.flat_p.cpu().numpy(), float(t["loss"]), float(t["bce"]), float(t["dice"])))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_full_width_bf16_step_under_sync_bn():
    """The benchmarked model (full-width UNet, bf16: recomputed stem, pool / head tails) through the data-parallel SyncBN
    path: both ranks end with bit-identical parameters (same global statistics, same all-reduced gradients) and their
    global-batch BCE / Dice match the single-process step on the concatenated batch at bf16 tolerance."""
    import unet_amd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unet_amd.UNet(1, 1, bilinear=True).to(memory_format=torch.channels_last).to(dev)
    st = unet_amd.TrainStepper(model, lr=1e-5, amp=True)
    im, mk = unet_amd.ellipse_batch(4, 64, seed=77)
    t = None
    for _ in range(2):
        t = st.step(im.to(dev), mk.to(dev))
    torch.cuda.synchronize()
    ref = (float(t["bce"]), float(t["dice"]))
    st.optimizer.close()
    res = _spawn(_fullwidth_syncbn_worker)
    assert (res[0][2] == res[1][2]).all(), "ranks diverged"
    for _, _, p, loss, bce, dice in res:
        import numpy as np
        assert np.isfinite(p).all() and np.isfinite(loss)
        assert abs(bce - ref[0]) <= 3e-2 * abs(ref[0]) and abs(dice - ref[1]) <= 3e-2 * abs(ref[1]), ((bce, dice), ref)
