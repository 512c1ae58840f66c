"""bench.py as the driver starts it for N > 1, rehearsed on the test GPU: `--gpus 2` with no torch.distributed environment
must start two ranks, run the weak headline + the global-batch-32 leg + the sustained leg with the gradient / loss-sum
collectives in every step, and print ONE JSON line with n_gpus == collective.world == 2.  gloo + --share-gpu on a one-GPU
box; RCCL ("nccl") when the machine has two devices."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


COMMON = ["--steps", "2", "--warmup", "1", "--size", "128", "--no-cpu-baseline", "--no-inference", "--sustained-seconds", "0.05"]


def test_one_rank_line_is_unchanged():
    out = _bench(["--gpus", "1"] + COMMON)
    assert out["n_gpus"] == 1 and "collective" not in out and "strong_gb32" not in out
    assert out["scaling"] == "weak" and out["sustained"]["steps"] >= 2
    assert out["roofline"]["bound"] == "mfma" and "double_conv_256_after_sustained" in out["kernels"]


def test_two_ranks_over_gloo_on_one_gpu():
    out = _bench(["--gpus", "2", "--backend", "gloo", "--share-gpu"] + COMMON)
    assert out["n_gpus"] == 2 and out["collective"]["world"] == 2 and out["collective"]["backend"] == "gloo"
    assert out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    sg = out["strong_gb32"]
    assert sg["global_batch"] == 32 and sg["per_gpu_batch"] == 16 and sg["ms_per_step"] > 0
    assert out["collective"]["exposed_allreduce_ms_per_step"] is not None
    assert out["sustained"]["images_per_sec"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs for an RCCL run")
def test_two_ranks_over_rccl():
    out = _bench(["--gpus", "2"] + COMMON)
    assert out["n_gpus"] == 2 and out["collective"]["world"] == 2 and out["collective"]["backend"] == "nccl"
    assert out["strong_gb32"]["global_batch"] == 32
