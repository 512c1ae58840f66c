"""bench.py --gpus N must start N ranks itself when no torch.distributed environment is present (missing #1).
Without a GPU the ranks stop at the "needs an MI355X" check -- after the launcher has given each of them its RANK /
WORLD_SIZE, which is what is checked here; the full 2-rank flow runs in tests/test_gpu_bench_ranks.py on the GPU box."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_flag_spawns_that_many_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check of the launcher (the GPU flow has its own test)")
    r = _run(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "rank 0/2" in r.stderr and "rank 1/2" in r.stderr, r.stderr[-2000:]


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_more_ranks_than_devices_is_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("machine has several GPUs")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "--share-gpu" in r.stderr
