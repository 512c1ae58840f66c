"""world_size-2 gloo tests (CPU) of the data-parallel plumbing in dp.py: bucketed asynchronous
all-reduce (sum / mean) of a flat gradient buffer in backward-ready order, and the loss-sum reducer that makes
Dice / BCE those of the GLOBAL batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


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
        import unet_amd.dp as dp
        torch.manual_seed(100 + rank)
        sizes = [5, 1300, 7, 64, 2048, 3, 900]
        slices, off = [], 0
        for n in sizes:
            slices.append((off, n))
            off += (n + 3) // 4 * 4
        flat = torch.zeros(off)
        mine = torch.randn(off)
        for average in (False, True):              # SUM is what the train step uses (globally normalised loss)
            sync = dp.BucketedGradSync(flat, slices, bucket_bytes=4 * 1500, average=average)
            assert len(sync.buckets) >= 3 and sync.buckets[0][0] == 0 and sync.buckets[-1][1] == off
            for step in range(2):                      # two steps: state must reset
                flat.copy_(mine * (step + 1))
                for i in [0, 2, 1, 3, 5, 4, 6]:        # not exactly monotonic, like real autograd hooks
                    sync.mark_ready(i)
                sync.wait()
                gathered = [torch.zeros(off) for _ in range(world)]
                dist.all_gather(gathered, mine * (step + 1))
                want = sum(gathered) / (world if average else 1)
                assert torch.allclose(flat, want, atol=1e-6), f"rank {rank} step {step} average {average}"
        # a parameter that never produced a gradient must not dead-lock the step
        flat.copy_(mine)
        sync.mark_ready(0)
        sync.wait()
        red = dp.make_sum_reducer()
        t = torch.tensor([1.0 + rank, 2.0, 3.0 * rank, 4.0])
        red(t)
        assert torch.allclose(t, torch.tensor([3.0, 4.0, 3.0, 8.0]))
        assert dp.world_size() == world
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


def test_bucketed_grad_sync_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def test_single_process_is_a_no_op():
    import unet_amd.dp as dp
    flat = torch.arange(16.0)
    sync = dp.BucketedGradSync(flat.clone(), [(0, 8), (8, 8)], bucket_bytes=16)
    sync.mark_ready(1)
    sync.mark_ready(0)
    sync.wait()
    assert torch.equal(sync.flat, flat)
    assert dp.make_sum_reducer() is None
