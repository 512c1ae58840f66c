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


def _bench(args, timeout=360):       # (below the GPU box's 420 s silence limit: a hung rank fails the test, not the run)
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
    assert out["n_gpus"] == 1 and "collective" not in out
    # (round 5: the fixed-global-batch leg runs at N = 1 too -- 32 images on the one GPU -- so that 1 -> N scaling at global batch 32
    # is the ratio of two strong_gb32 figures measured the same way)
    sg = out["strong_gb32"]
    assert sg["global_batch"] == 32 and sg["per_gpu_batch"] == 32 and sg["images_per_sec"] > 0 and sg["exposed_allreduce_ms_per_step"] is None
    assert out["scaling"] == "weak" and out["sustained"]["steps"] >= 2
    assert out["roofline"]["bound"] == "mfma" and "double_conv_256_after_sustained" in out["kernels"]
    # every launch of the forward / backward-data kernel beside the plain family, and config 3's per-GPU batch on one GPU
    assert 0 < out["roofline"]["all_launches_frac"] <= 1 and out["roofline"]["all_launches"]["launches_per_step"] == 34
    b4 = out["per_gpu_batch4"]
    assert b4["per_gpu_batch"] == 4 and b4["ms_per_step"] > 0 and 0 < b4["vs_batch8_images_per_sec"] < 2
    assert b4["double_conv_256_in_step"]["all_six"]["tflops"] > 0


def test_bf16x3_roofline_is_quoted_on_the_pipe_that_runs_it():
    """fp32 tensors with bf16x3 split products: three bf16 MFMA products per algorithmic one, quoted against the bf16 peak -- the
    fraction must be a fraction (round 3 divided by the fp32 peak: 1.59)."""
    out = _bench(["--gpus", "1", "--fp32", "--bf16x3", "--no-sustained", "--no-b4-leg"] + COMMON)
    r = out["roofline"]
    assert r["peak"] == 2500.0 and 0 < r["frac"] <= 1.0, r
    assert abs(r["pipe_tflops"] - 3 * r["achieved"]) < 0.5, r


def test_four_ranks_over_gloo_on_one_gpu_share_the_config_keys_of_the_one_rank_line():
    """Round 5 (review item 8): the N-rank line and the one-rank line must carry the same `config` keys and the same
    `strong_gb32` keys (the 1 -> N scaling at global batch 32 is the ratio of the two strong_gb32 figures); four ranks is what the
    box's process limit allows on one card (the driver's run is eight, one per GPU, over RCCL)."""
    one = _bench(["--gpus", "1"] + COMMON)
    four = _bench(["--gpus", "4", "--backend", "gloo", "--share-gpu"] + COMMON)
    assert four["n_gpus"] == 4 and four["collective"]["world"] == 4 and four["config"]["global_batch"] == 32
    assert set(one["config"]) == set(four["config"])
    assert four["strong_gb32"]["per_gpu_batch"] == 8 and one["strong_gb32"]["per_gpu_batch"] == 32
    assert set(one["strong_gb32"]) == set(four["strong_gb32"])
    col = four["collective"]
    for key in ("backend", "world", "rccl_version", "per_gpu_batch", "global_batch", "grad_bytes", "grad_buckets", "bucket_bytes",
                "allreduce", "sync_bn", "exposed_allreduce_ms_per_step"):
        assert key in col, key
    assert sum(col["bucket_bytes"]) == col["grad_bytes"] and len(col["bucket_bytes"]) == col["grad_buckets"]


def test_two_ranks_over_gloo_on_one_gpu():
    out = _bench(["--gpus", "2", "--backend", "gloo", "--share-gpu"] + COMMON)
    assert out["n_gpus"] == 2 and out["collective"]["world"] == 2 and out["collective"]["backend"] == "gloo"
    assert out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    sg = out["strong_gb32"]
    assert sg["global_batch"] == 32 and sg["per_gpu_batch"] == 16 and sg["ms_per_step"] > 0
    assert sg["double_conv_256_in_step"]["all_six"]["tflops"] > 0           # the kernel figure at the strong leg's per-GPU batch
    assert out["collective"]["exposed_allreduce_ms_per_step"] is not None
    assert out["sustained"]["images_per_sec"] > 0


def test_config5_recipe_through_the_two_rank_launcher():
    """BASELINE config 5 exactly as the multi-GPU driver would start it (transposed-conv upsample, fp32, connected_component_loss,
    global batch 16), at two ranks over gloo on the test GPU: the line names the config, carries the convT model's gradient
    payload (124.15 MB) and the exposed all-reduce time.  Unmeasured over RCCL on this pool (one-GPU boxes)."""
    out = _bench(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--convt", "--fp32", "--cc-loss", "--global-batch", "16",
                  "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-inference", "--sustained-seconds", "0.05"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["dtype"] == "f32"
    assert out["config"]["baseline_config"].startswith("configs[4]"), out["config"]
    assert out["config"]["global_batch"] == 16 and out["config"]["per_gpu_batch"] == 8
    assert "bilinear=False" in out["config"]["workload"] and "512x512" in out["config"]["workload"]
    c = out["collective"]
    assert c["world"] == 2 and abs(c["grad_bytes"] / 1e6 - 124.15) < 0.01, c
    assert c["exposed_allreduce_ms_per_step"] is not None
    assert out["loss"] == out["loss"]                                        # finite: the cc term is a Python float added to the value


def test_config3_strong_leg_with_sync_bn_through_the_two_rank_launcher():
    """BASELINE config 3's workload (global batch 32 over the ranks, bf16, bilinear) with global-batch BatchNorm statistics, at two
    ranks over gloo on the test GPU: names the config, 69.05 MB of gradients per step, SyncBN stated in the line."""
    out = _bench(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--global-batch", "32", "--sync-bn",
                  "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-inference", "--sustained-seconds", "0.05"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["dtype"] == "bf16"
    assert out["config"]["baseline_config"].startswith("configs[2]"), out["config"]
    assert out["config"]["global_batch"] == 32 and out["config"]["per_gpu_batch"] == 16
    assert "SyncBN" in out["config"]["bn"]
    c = out["collective"]
    assert c["world"] == 2 and abs(c["grad_bytes"] / 1e6 - 69.05) < 0.01, c
    assert c["exposed_allreduce_ms_per_step"] is not None
    assert out["kernels"]["double_conv_256_in_step"]["all_six"]["tflops"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs for an RCCL run")
def test_two_ranks_over_rccl():
    out = _bench(["--gpus", "2"] + COMMON)
    assert out["n_gpus"] == 2 and out["collective"]["world"] == 2 and out["collective"]["backend"] == "nccl"
    assert out["strong_gb32"]["global_batch"] == 32
