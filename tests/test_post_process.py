"""Mask post-processing.  CPU: the scipy oracle on hand-derived cases (parity with OpenCV is
UNPINNED -- it is not installed).  GPU: csrc/post_process.hip against the oracle on random blob masks."""
import numpy as np
import pytest
import torch

from oracle import post_process_ref as R


def _blobs(rng, H, W, n, rmax):
    yy, xx = np.mgrid[0:H, 0:W]
    m = np.ones((H, W), np.uint8)
    for _ in range(n):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        ry, rx = rng.uniform(2, rmax, 2)
        m[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = 2
    for _ in range(n):                                        # holes / nested background
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        r = rng.uniform(1, rmax / 3)
        m[((yy - cy) / r) ** 2 + ((xx - cx) / r) ** 2 <= 1] = rng.integers(0, 2)
    noise = rng.random((H, W))
    m[noise < 0.01] = 2                                       # specks the opening must remove
    m[noise > 0.995] = 0
    return m


def test_oracle_known_answers():
    m = np.ones((12, 12), np.uint8)
    m[2:10, 2:10] = 2
    m[4:8, 4:8] = 1                                           # hole
    m[5, 5] = 2                                               # island inside the hole
    out = R.remove_internal_regions(m)
    assert (out[2:10, 2:10] == 2).all() and (out[:2] == 1).all()
    m2 = np.ones((12, 12), np.uint8)
    m2[2:10, 2:10] = 2
    m2[4:8, 4:8] = 0
    m2[4:8, 9] = 0                                            # ... the hole now leaks to the outside through a channel
    m2[4:8, 8] = 0
    m2[4:8, 10:12] = 0
    assert (R.remove_internal_regions(m2)[4:8, 4:8] == 0).all()
    # diagonal wall: 8-connected foreground ring encloses its interior (background is only 4-connected)
    ring = np.ones((7, 7), np.uint8)
    for (y, x) in [(1, 3), (2, 2), (3, 1), (4, 2), (5, 3), (4, 4), (3, 5), (2, 4)]:
        ring[y, x] = 2
    assert R.remove_internal_regions(ring)[3, 3] == 2
    # opening + area filter + {0,2} output
    big = np.ones((40, 40), np.uint8)
    big[5:30, 5:30] = 2                                       # 625 px: kept at min_area 500
    big[35, 35] = 2                                           # speck: removed by the opening
    big[32:36, 2:6] = 2                                       # 16 px: removed by the area filter
    out = R.postprocess_mask(big, min_area=500)
    assert set(np.unique(out)) == {0, 2} and (out[5:30, 5:30] == 2).all() and out[35, 35] == 0 and (out[32:36, 2:6] == 0).all()
    assert (out[0] == 0).all()                                # the class-1 background is zeroed too (post_process.py:84-86)
    # the opening does not erode at the image border (OpenCV's border rule)
    edge = np.ones((20, 20), np.uint8)
    edge[0:12, 0:12] = 2
    assert (R.postprocess_mask(edge, min_area=100)[0:12, 0:12] == 2).all()


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,min_area", [(2, 64, 64, 60), (3, 97, 131, 150), (2, 512, 512, 15000), (1, 33, 17, 1)])
def test_device_postprocess_matches_oracle(B, H, W, min_area):
    import unet_amd
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    rng = np.random.default_rng(H * W + B)
    masks = np.stack([_blobs(rng, H, W, 6 + H // 32, max(6, min(H, W) / 4)) for _ in range(B)])
    ref = np.stack([R.postprocess_mask(m, min_area=min_area) for m in masks])
    got = unet_amd.postprocess_mask(torch.from_numpy(masks).cuda(), min_area=min_area)
    assert got.dtype == torch.uint8 and tuple(got.shape) == masks.shape
    assert np.array_equal(got.cpu().numpy(), ref)
    # numpy in -> numpy out, single image, int64
    one = unet_amd.postprocess_mask(masks[0].astype(np.int64), min_area=min_area)
    assert isinstance(one, np.ndarray) and one.dtype == np.int64 and np.array_equal(one, ref[0])
    fill = unet_amd.remove_internal_regions(torch.from_numpy(masks).cuda())
    assert np.array_equal(fill.cpu().numpy(), np.stack([R.remove_internal_regions(m) for m in masks]))
    assert np.array_equal(unet_amd.postprocess_mask(torch.from_numpy(masks).cuda(), min_area=min_area, morph_kernel_size=5).cpu().numpy(),
                          np.stack([R.postprocess_mask(m, min_area=min_area, morph_kernel_size=5) for m in masks]))


@pytest.mark.gpu
def test_evaluate_with_postprocess():
    import unet_amd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unet_amd.UNet_T(1, 3, bilinear=True).to(dev)
    im, mk = unet_amd.ellipse_batch(4, 64, seed=3)
    d0, d1, dmin = unet_amd.evaluate(model, [{"image": im, "mask": mk}], dev, amp=False, postprocess=True)
    assert 0.0 <= float(d0) <= 1.0 and 0.0 <= float(d1) <= 1.0
    model1 = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    d0, d1, dmin = unet_amd.evaluate(model1, [{"image": im, "mask": mk}], dev, amp=False, postprocess=True)
    assert float(dmin) <= float(d0) + 1e-6
