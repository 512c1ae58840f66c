"""Host side of the device input stage: the numpy oracle (oracle/data_prep_ref.py) against the reference's own dataset
items (fixture G12, scale 1.0: rotation with expand, /255 rule, label remap), and BasicDataset.raw_item / collate_raw,
which stop before any per-pixel arithmetic."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_data_loading_cpu import _write_tree


def test_oracle_matches_reference_items():
    from oracle import data_prep_ref as R
    r = load_golden("g12_data_loading")
    names = sorted({k.split(".")[1] for k in r.keys() if k.startswith("raw.")})
    for n in names:
        for rot in range(4):
            img, mask = R.prepare_item(r[f"raw.{n}.img"], r[f"raw.{n}.mask"], rot)
            assert np.array_equal(img, r[f"s1.0.{n}.r{rot}.image"]) and img.dtype == np.float32, (n, rot)
            assert np.array_equal(mask, r[f"s1.0.{n}.r{rot}.mask"]) and mask.dtype == np.int64, (n, rot)


def test_oracle_keeps_binary_images_undivided():
    from oracle import data_prep_ref as R
    img = np.array([[0, 1], [1, 0]], np.uint8)
    out, _ = R.prepare_item(img, np.zeros((2, 2), np.uint8))
    assert np.array_equal(out[0], img.astype(np.float32))             # no value above 1: data_loading.py:86 leaves it


@pytest.mark.parametrize("scale", [1.0, 0.5])
def test_raw_items_reproduce_the_reference_items(tmp_path, scale):
    """raw_item + the oracle's arithmetic == the reference's item, for both the decode-only form (scale 1) and the
    host-rescaled form (scale < 1, turns already applied)."""
    from oracle import data_prep_ref as R
    from unet_amd.utils.data_loading import BasicDataset
    r = load_golden("g12_data_loading")
    names = _write_tree(tmp_path, r)
    ds = BasicDataset(str(tmp_path / "imgs"), str(tmp_path / "masks"), scale, augment=True)
    for n in names:
        base = ds.ids.index(n) * 4
        for rot in range(4):
            raw = ds.raw_item(base + rot)
            assert raw["image_u8"].dtype == np.uint8 and raw["mask_u8"].dtype == np.uint8
            assert raw["turns"] == (rot if scale == 1.0 else 0)
            img, mask = R.prepare_item(raw["image_u8"], raw["mask_u8"], raw["turns"])
            assert np.array_equal(img, r[f"s{scale}.{n}.r{rot}.image"]), (n, rot)
            assert np.array_equal(mask, r[f"s{scale}.{n}.r{rot}.mask"]), (n, rot)


def test_collate_raw_shapes_and_refusals(tmp_path):
    from unet_amd.utils.data_loading import BasicDataset, collate_raw
    r = load_golden("g12_data_loading")
    _write_tree(tmp_path, r)
    ds = BasicDataset(str(tmp_path / "imgs"), str(tmp_path / "masks"), 1.0, augment=True)
    base = ds.ids.index("a01") * 4
    b = collate_raw([ds.raw_item(base + 1), ds.raw_item(base + 3)], pin=False)
    h, w = r["raw.a01.img"].shape
    assert tuple(b["image_u8"].shape) == (2, h, w, 1) and tuple(b["mask_u8"].shape) == (2, h, w)
    assert b["turns"].tolist() == [1, 3] and b["image_u8"].dtype == torch.uint8
    with pytest.raises(ValueError):
        collate_raw([ds.raw_item(base + 0), ds.raw_item(base + 1)], pin=False)        # 20x24 rotated by 0 and 1 turns
    other = ds.ids.index("b02") * 4
    with pytest.raises(ValueError):
        collate_raw([ds.raw_item(base), ds.raw_item(other)], pin=False)               # two decoded sizes


def test_device_stage_has_no_cpu_fallback():
    from unet_amd.utils.data_loading import prepare_batch_device
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises((RuntimeError, AssertionError)):
        prepare_batch_device(torch.zeros(1, 4, 4, 1, dtype=torch.uint8), torch.zeros(1, 4, 4, dtype=torch.uint8), device="cpu")
