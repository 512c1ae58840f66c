"""Checkpoint wire format (train.py:208-216 / 275-280, predict.py:106-109): state_dict + 'mask_values', loadable into
stock torch modules with the reference's key layout (golden fixture G8 holds the reference's keys and shapes)."""
import numpy as np
import torch

from conftest import load_golden


def test_checkpoint_format_and_reference_keys(tmp_path):
    import unet_amd
    r = load_golden("g8_unet_t_bilinear")
    ref_keys = [k[len("sd0."):] for k in r.keys() if k.startswith("sd0.")]
    model = unet_amd.UNet_T(1, 1, bilinear=True)
    path = unet_amd.save_checkpoint(model, str(tmp_path / "m.pth"), mask_values=[0, 128, 255, 0, 128, 255])
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert list(raw.keys()) == ref_keys + ["mask_values"]
    for k in ref_keys:
        assert tuple(raw[k].shape) == tuple(np.asarray(r["sd0." + k]).shape), k
    # a checkpoint written the reference's way (plain dict incl. mask_values) loads back
    ref_sd = {k: torch.from_numpy(np.asarray(r["sd0." + k]).copy()) for k in ref_keys}
    ref_sd["mask_values"] = [0, 128, 255]
    torch.save(ref_sd, str(tmp_path / "ref.pth"))
    other = unet_amd.UNet_T(1, 1, bilinear=True)
    assert unet_amd.load_checkpoint(other, str(tmp_path / "ref.pth")) == [0, 128, 255]
    for k, v in other.state_dict().items():
        assert torch.equal(v, ref_sd[k]), k


def test_preprocess_image_matches_reference_rules():
    import unet_amd
    from PIL import Image
    a = (np.arange(6 * 8).reshape(6, 8) * 5).astype(np.uint8)
    out = unet_amd.preprocess_image(Image.fromarray(a, mode="L"))
    assert out.shape == (1, 6, 8) and out.dtype == np.float32
    assert np.allclose(out[0], a.astype(np.float32) / 255.0)
    rgb = np.stack([a, a, a], -1)
    out = unet_amd.preprocess_image(Image.fromarray(rgb, mode="RGB"))
    assert out.shape == (3, 6, 8)
    tiny = np.zeros((4, 4), np.uint8); tiny[0, 0] = 1          # no value above 1 -> NOT divided (data_loading.py:87-88)
    out = unet_amd.preprocess_image(Image.fromarray(tiny, mode="L"))
    assert out.max() == 1
