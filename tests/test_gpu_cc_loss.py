"""connected_component_loss on the device (uh_cc_loss_device: union-find components, contourArea as a sum over 2x2 blocks)
against the host border follower (uh_cc_loss_host) -- two independent algorithms for the same OpenCV calls
(utils/connected_component_loss.py:28-56) -- and against hand-derived known answers."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    return torch.device("cuda:0")


def _host(masks, ed, ma):
    from unet_amd._lib import LIB
    m = np.ascontiguousarray(masks.astype(np.uint8))
    out = (ctypes.c_double * 2)()
    LIB.call("uh_cc_loss_host", m.ctypes.data, m.shape[0], m.shape[1], m.shape[2], ed, ma, ctypes.cast(out, ctypes.c_void_p))
    return out[0], out[1]


def _device(masks, ed, ma):
    from unet_amd._lib import LIB
    dev = _dev()
    m = torch.from_numpy(np.ascontiguousarray(masks.astype(np.uint8))).to(dev)
    B, H, W = m.shape
    nbytes = LIB.query("uh_cc_loss_ws_bytes", B, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    res = torch.empty(2, dtype=torch.float64, device=dev)
    LIB.call("uh_cc_loss_device", m.data_ptr(), B, H, W, ed, ma, ws.data_ptr(), nbytes, res.data_ptr(),
             torch.cuda.current_stream().cuda_stream)
    r = res.cpu()
    return float(r[0]), float(r[1])


def _blob_masks(rng, B, H, W, kind):
    if kind == "noise":
        return rng.random((B, H, W)) < rng.uniform(0.2, 0.8)
    if kind == "sparse":
        return rng.random((B, H, W)) < 0.03
    # smooth blobs with holes and nested islands
    from scipy import ndimage as ndi
    f = ndi.gaussian_filter(rng.random((B, H, W)), (0, 3, 3))
    m = f > np.quantile(f, 0.55)
    m &= ~(ndi.gaussian_filter(rng.random((B, H, W)), (0, 2, 2)) > 0.53)        # punch holes
    m |= rng.random((B, H, W)) < 0.01                                           # specks, some inside the holes
    return m


@pytest.mark.parametrize("kind", ["noise", "sparse", "blobs"])
@pytest.mark.parametrize("B,H,W", [(3, 37, 53), (2, 64, 64), (1, 5, 9), (4, 128, 96), (2, 512, 512)])
def test_device_equals_host_border_follower(kind, B, H, W):
    rng = np.random.default_rng(B * 1000 + H + W + len(kind))
    m = _blob_masks(rng, B, H, W, kind)
    for ed, ma in [(5, 7), (50, 1000), (16, 40)]:
        h, d = _host(m, ed, ma), _device(m, ed, ma)
        assert h[1] == d[1], (h, d)                      # number of external contours
        assert abs(h[0] - d[0]) <= 1e-12 * max(1.0, abs(h[0])), (h, d)


def test_known_answers_on_the_device():
    """single pixel: area 0; filled w x h rectangle: (w-1)(h-1); ring: the area of its outer border, the island in its
    hole is not an external contour; diagonal chain: area 0; L of three pixels: 1/2."""
    H = W = 40
    ma, ed = 10, 4

    def loss(m):
        return _device(m[None], ed, ma)

    m = np.zeros((H, W), bool); m[20, 20] = True
    assert loss(m) == (1.0, 1.0)                                            # area 0 < min_area: 1 - 0/10
    m = np.zeros((H, W), bool); m[10:14, 10:16] = True                      # 4 x 6 rectangle: area 3*5 = 15, centre far from edges
    assert loss(m) == (0.0, 1.0)
    m = np.zeros((H, W), bool); m[10:13, 10:13] = True                      # 3 x 3: area 4 -> 1 - 4/10
    p, c = loss(m); assert c == 1.0 and abs(p - 0.6) < 1e-15
    m = np.zeros((H, W), bool); m[8:20, 8:20] = True; m[10:18, 10:18] = False; m[13:15, 13:15] = True      # ring + island
    assert loss(m) == (0.0, 1.0)                                            # one external contour, area 11*11 = 121
    m = np.zeros((H, W), bool)
    for i in range(6):
        m[5 + i, 5 + i] = True                                             # diagonal chain: one component, area 0
    assert loss(m) == (1.0, 1.0)
    m = np.zeros((H, W), bool); m[20, 20] = m[20, 21] = m[21, 20] = True    # L: area 1/2 -> 1 - 0.05
    p, c = loss(m); assert c == 1.0 and abs(p - 0.95) < 1e-15
    m = np.zeros((H, W), bool); m[0:6, 0:6] = True                          # area 25 >= 10; bbox centre (3,3): d = 3 < 4
    p, c = loss(m); assert c == 1.0 and abs(p - 0.25) < 1e-15


def test_python_surface_uses_the_device_path():
    from unet_amd.utils.connected_component_loss import connected_component_loss
    dev = _dev()
    rng = np.random.default_rng(7)
    probs = torch.from_numpy(rng.random((2, 96, 80)).astype(np.float32))
    probs = torch.nn.functional.avg_pool2d(probs[None], 5, 1, 2)[0] * 1.2       # blobs
    a = connected_component_loss(probs.to(dev), edge_distance=10, min_area=30, penalty_weight=0.1)
    b = connected_component_loss(probs, edge_distance=10, min_area=30, penalty_weight=0.1)
    assert isinstance(a, float) and abs(a - b) < 1e-12
