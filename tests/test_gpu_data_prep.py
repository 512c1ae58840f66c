"""Device stage of the input pipeline (csrc/data_prep.hip, uh_batch_prepare) through the C ABI: bit-exact against the
reference's own dataset items (fixture G12) and against the numpy oracle on multi-tile / multi-channel / mixed-turn
batches; the per-image /255 rule; bf16 output = torch's rounding of the fp32 values; the result feeds the model as is."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _prep(img, mask, turns=None, dtype=torch.float32):
    from unet_amd.utils.data_loading import prepare_batch_device
    out = prepare_batch_device(torch.from_numpy(np.ascontiguousarray(img)), torch.from_numpy(np.ascontiguousarray(mask)),
                               turns, device="cuda", dtype=dtype)
    torch.cuda.synchronize()
    return out


def test_g12_items_bit_exact():
    r = load_golden("g12_data_loading")
    names = sorted({k.split(".")[1] for k in r.keys() if k.startswith("raw.")})
    for n in names:
        img, mask = np.asarray(r[f"raw.{n}.img"]), np.asarray(r[f"raw.{n}.mask"])
        for rots in ((0, 2), (1, 3)):                      # non-square files: one parity per batch
            out = _prep(np.stack([img, img]), np.stack([mask, mask]), list(rots))
            assert out["image"].dtype == torch.float32 and out["mask"].dtype == torch.int64
            for b, rot in enumerate(rots):
                want_i, want_m = r[f"s1.0.{n}.r{rot}.image"], r[f"s1.0.{n}.r{rot}.mask"]
                assert tuple(out["image"][b].shape) == want_i.shape
                assert np.array_equal(out["image"][b].cpu().numpy(), want_i), (n, rot)
                assert np.array_equal(out["mask"][b].cpu().numpy(), want_m), (n, rot)


@pytest.mark.parametrize("H,W,C", [(512, 512, 1), (130, 70, 3), (64, 64, 4), (1, 200, 1), (257, 65, 2)])
def test_against_numpy_oracle(H, W, C):
    from oracle import data_prep_ref as R
    rng = np.random.default_rng(H * 1000 + W + C)
    B = 5
    img = rng.integers(0, 256, size=(B, H, W, C), dtype=np.uint8)
    img[3] = rng.integers(0, 2, size=(H, W, C), dtype=np.uint8)          # a 0/1 image: NOT divided (data_loading.py:86)
    mask = rng.choice(np.array([0, 128, 255, 7, 254], np.uint8), size=(B, H, W))
    if H == W:
        turn_sets = ([0, 1, 2, 3, 1], None)
    else:
        turn_sets = ([0, 2, 2, 0, 2], [1, 3, 3, 1, 1], None)
    for turns in turn_sets:
        want_i, want_m = R.prepare_batch([im if C > 1 else im[..., 0] for im in img], mask, turns)
        out = _prep(img, mask, turns)
        assert np.array_equal(out["image"].cpu().numpy(), want_i), turns
        assert np.array_equal(out["mask"].cpu().numpy(), want_m), turns
        assert out["image"].is_contiguous(memory_format=torch.channels_last) or C == 1
        bf = _prep(img, mask, turns, dtype=torch.bfloat16)
        assert torch.equal(bf["image"].cpu(), torch.from_numpy(want_i).to(torch.bfloat16)), turns


def test_feeds_the_train_step():
    import unet_amd
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(2, 64, 64, 1), dtype=np.uint8)
    mask = rng.choice(np.array([0, 128, 255], np.uint8), size=(2, 64, 64))
    batch = _prep(img, mask, [1, 2])
    torch.manual_seed(0)
    model = unet_amd.UNet_T(1, 1, bilinear=True).cuda()
    stepper = unet_amd.TrainStepper(model, amp=False)
    t = stepper.step(batch["image"], batch["mask"])
    # the same batch prepared by the host path
    from oracle import data_prep_ref as R
    hi, hm = R.prepare_batch([im[..., 0] for im in img], mask, [1, 2])
    torch.manual_seed(0)
    model2 = unet_amd.UNet_T(1, 1, bilinear=True).cuda()
    st2 = unet_amd.TrainStepper(model2, amp=False)
    t2 = st2.step(torch.from_numpy(hi).cuda(), torch.from_numpy(hm).cuda())
    assert torch.equal(t["logits"], t2["logits"]) and float(t["loss"]) == float(t2["loss"])


def test_bad_arguments_raise():
    from unet_amd.utils.data_loading import prepare_batch_device
    u8 = torch.zeros(2, 8, 12, 1, dtype=torch.uint8)
    with pytest.raises(ValueError):
        prepare_batch_device(u8, torch.zeros(2, 8, 12, dtype=torch.uint8), [0, 1], device="cuda")      # mixed parity, non-square
    with pytest.raises(TypeError):
        prepare_batch_device(u8.float(), torch.zeros(2, 8, 12, dtype=torch.uint8), device="cuda")
