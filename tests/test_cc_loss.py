"""connected_component_loss: host analysis in libunet_hip.so (uh_cc_loss_host) against
hand-derived known answers and against the independent numpy/scipy restatement in oracle/cc_loss_ref.py.
Parity with OpenCV itself is UNPINNED (OpenCV is not installed)."""
import numpy as np
import pytest
import torch

from oracle import cc_loss_ref as R


def cc(mask, **kw):
    from unet_amd.utils.connected_component_loss import connected_component_loss
    return connected_component_loss(torch.as_tensor(mask, dtype=torch.float32), **kw)


def blank(h=40, w=50):
    return np.zeros((1, h, w), dtype=np.float32)


def test_single_pixel_has_zero_area():
    m = blank(); m[0, 7, 9] = 1
    assert cc(m, min_area=10, edge_distance=3, penalty_weight=1.0) == pytest.approx(1.0)


@pytest.mark.parametrize("w,h", [(7, 4), (2, 2), (12, 1), (1, 9), (20, 15)])
def test_filled_rectangle_area_is_w_minus_1_times_h_minus_1(w, h):
    m = blank(); m[0, 10:10 + h, 12:12 + w] = 1
    area = (w - 1) * (h - 1)
    got = cc(m, min_area=1000, edge_distance=1, penalty_weight=1.0)
    assert got == pytest.approx(1.0 - area / 1000.0)


def test_hole_is_ignored_and_nested_blob_is_not_external():
    m = blank(); m[0, 5:25, 5:30] = 1
    m[0, 9:21, 9:26] = 0                     # hole
    m[0, 13:17, 14:20] = 1                   # island inside the hole: not RETR_EXTERNAL
    area = 24 * 19
    assert cc(m, min_area=10000, edge_distance=1, penalty_weight=1.0) == pytest.approx(1.0 - area / 10000.0)


def test_diagonal_line_and_two_blobs():
    m = blank()
    for i in range(6):
        m[0, 3 + i, 4 + i] = 1               # 8-connected diagonal: degenerate chain, area 0
    assert cc(m, min_area=5, edge_distance=1, penalty_weight=1.0) == pytest.approx(1.0)
    m[0, 20:24, 30:36] = 1                   # second component, area 15
    assert cc(m, min_area=30, edge_distance=1, penalty_weight=1.0) == pytest.approx(1.0 + 0.5)


def test_plus_shape_diamond_area():
    m = blank(); m[0, 10, 9:12] = 1; m[0, 9:12, 10] = 1
    assert cc(m, min_area=4, edge_distance=1, penalty_weight=1.0) == pytest.approx(1.0 - 2.0 / 4.0)


def test_edge_distance_penalty_and_batch_mean_and_weight():
    m = np.zeros((2, 60, 80), dtype=np.float32)
    m[0, 2:22, 30:60] = 1                    # area 19*29 = 551 >= 100; bbox x=30,y=2,w=30,h=20 -> centre (45, 12)
    # d = min(45, 80-45, 12, 60-12) = 12 < 50 -> 1 - 12/50
    m[1, 25:35, 35:45] = 1                   # area 81 < 100 -> 1 - 81/100
    want = ((1 - 12 / 50) + (1 - 81 / 100)) / 2 * 0.1
    assert cc(m, edge_distance=50, min_area=100, penalty_weight=0.1) == pytest.approx(want)
    assert cc(m * 0.5, edge_distance=50, min_area=100) == 0.0        # p > 0.5 is strict


def test_against_independent_restatement_on_random_blobs():
    rng = np.random.default_rng(0)
    for trial in range(12):
        H, W = 48, 64
        img = np.zeros((2, H, W), dtype=np.float32)
        for b in range(2):
            for _ in range(rng.integers(1, 5)):
                cy, cx = rng.integers(0, H), rng.integers(0, W)
                ry, rx = rng.integers(2, 12), rng.integers(2, 14)
                yy, xx = np.ogrid[:H, :W]
                img[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 1.0    # filled ellipses (convex blobs)
        got = cc(img, edge_distance=10, min_area=60, penalty_weight=1.0)
        want = R.connected_component_loss(img, edge_distance=10, min_area=60, penalty_weight=1.0)
        assert got == pytest.approx(want, abs=1e-9), trial
