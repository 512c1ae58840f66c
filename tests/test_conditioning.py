"""CPU evidence for the gradient tolerances used by the GPU parity tests: stock PyTorch fp32 gradients of the
randomly initialised UNet move by ~1e-2 (relative L2) under 1e-7 relative perturbations of the conv weights,
i.e. under any change of fp32 summation order.  Logits stay within ~1e-5."""
import torch

from oracle import step_ref as S
from oracle import unet_ref as U


def test_whole_network_gradients_are_ill_conditioned_in_fp32():
    st = U.init_state(1, 1, True, widths=(16, 32, 64, 128, 256), seed=0)
    g = torch.Generator().manual_seed(1)
    images = torch.rand(2, 1, 64, 64, generator=g)
    masks = torch.randint(0, 3, (2, 64, 64), generator=g)
    st64 = {k: (v.double() if v.is_floating_point() else v) for k, v in st.items()}
    _, _, i64 = S.train_step(st64, None, images.double(), masks, n_classes=1, bilinear=True)
    worst = 0.0
    for trial in range(3):
        torch.manual_seed(trial)
        stp = {k: (v * (1 + (torch.rand_like(v) - 0.5) * 2e-7) if v.is_floating_point() and v.dim() == 4 else v)
               for k, v in st.items()}
        _, _, i32 = S.train_step(stp, None, images, masks, n_classes=1, bilinear=True)
        lerr = float((i32["logits"].double() - i64["logits"]).norm() / i64["logits"].norm())
        assert lerr < 1e-4
        for k in i64["grads"]:
            e = float((i32["grads"][k].double() - i64["grads"][k]).norm() / i64["grads"][k].norm())
            worst = max(worst, e)
    # fp32 round-off alone moves some gradient tensor by more than 1e-3 -- the north-star's 1e-3 applies to the
    # forward / loss values and to op-level results, not to whole-network gradients
    assert 1e-4 < worst < 5e-2, worst
