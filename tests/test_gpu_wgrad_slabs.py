"""Backward-weights on a REAL train step: the default 16-bit (block-scaled fp16) partial sums against fp32 partial sums
(UH_WGRAD_SLAB_F32=1, read once per process -- hence two child processes) on the same weights and the same batch of synthetic
ellipse masks, whose gradient has the foreground / background structure ADVICE r4 asked about.  Yardstick: the rms relative error
of rounding every element of the total to bf16 once, 2^-9 / sqrt(3) = 1.1e-3, which is what the reference's autocast backward
does to the filter gradient (train.py:116)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import unet_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = unet_amd.UNet(1, 1, bilinear=True).to(dev)
st = unet_amd.TrainStepper(model, lr=1e-4, amp=True)
im, mk = unet_amd.ellipse_batch(4, 256, seed=101)
st.step(im.to(dev), mk.to(dev))
torch.cuda.synchronize()
g = {k: st.optimizer.grad_of(p).detach().float().cpu().clone() for k, p in model.named_parameters() if p.dim() == 4 and p.shape[-1] == 3}
torch.save(g, sys.argv[1])
"""


def _grads(tmp_path, tag, env_extra):
    out = str(tmp_path / f"grads_{tag}.pt")
    env = dict(os.environ, **env_extra)
    env.pop("UH_LIB_PATH", None)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, out], env=env, capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stdout + r.stderr
    return torch.load(out)


def test_16_bit_partial_sums_on_a_real_train_step(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    half = _grads(tmp_path, "f16", {"UH_WGRAD_SLAB_F32": "0"})
    full = _grads(tmp_path, "f32", {"UH_WGRAD_SLAB_F32": "1"})
    worst = 0.0
    for k, ref in full.items():
        if k == "inc.double_conv.0.weight":
            continue                                       # the stem's filter gradient never had 16-bit partial sums
        ref = ref.double()
        rel = float((half[k].double() - ref).norm() / ref.norm().clamp_min(1e-300))
        rounding = float((ref.float().bfloat16().double() - ref).norm() / ref.norm().clamp_min(1e-300))
        worst = max(worst, rel / rounding)
        assert rel <= 0.5 * rounding, (k, rel, rounding)
    print(f"largest (16-bit partials - fp32 partials) / (bf16 rounding of the total) over the 3x3 layers: {worst:.3f}")
