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


# ------------------------------------------------------------------------------------------------------------------
# Independent check: the device kernel against the numpy / scipy restatement in oracle/cc_loss_ref.py
# (Moore border tracing + shoelace; a different algorithm from both csrc/ implementations), and OpenCV's documented corner
# cases derived by hand.  Still PARITY UNPINNED (OpenCV is not installed): this removes the self-comparison only.
# ------------------------------------------------------------------------------------------------------------------
def _oracle(masks, ed, ma):
    from oracle import cc_loss_ref as R
    # penalty_weight = 1  ->  (sum of penalties) / B, which is what the C ABI reports in result[0]
    return R.connected_component_loss(masks.astype(np.float32), edge_distance=ed, min_area=ma, penalty_weight=1.0)


@pytest.mark.parametrize("kind", ["noise", "sparse", "blobs"])
@pytest.mark.parametrize("B,H,W", [(3, 37, 53), (2, 64, 64), (1, 5, 9), (2, 128, 96)])
def test_device_equals_numpy_oracle(kind, B, H, W):
    from oracle import cc_loss_ref as R
    rng = np.random.default_rng(B * 77 + H + 3 * W + len(kind))
    m = _blob_masks(rng, B, H, W, kind)
    for ed, ma in [(5, 7), (50, 1000), (16, 40)]:
        d = _device(m, ed, ma)
        want = _oracle(m, ed, ma)
        ncont = sum(len(R.external_components(m[b])) for b in range(B))
        assert d[1] == ncont, (d, ncont)
        assert abs(d[0] - want) <= 1e-9 * max(1.0, abs(want)), (d, want)


def test_python_surface_equals_numpy_oracle_on_probabilities():
    from oracle import cc_loss_ref as R
    from unet_amd.utils.connected_component_loss import connected_component_loss
    dev = _dev()
    rng = np.random.default_rng(21)
    probs = torch.from_numpy(rng.random((3, 80, 112)).astype(np.float32))
    probs = torch.nn.functional.avg_pool2d(probs[None], 7, 1, 3)[0] * 1.15
    got = connected_component_loss(probs.to(dev), edge_distance=12, min_area=45, penalty_weight=0.1)
    want = R.connected_component_loss(probs.numpy(), edge_distance=12, min_area=45, penalty_weight=0.1)
    assert abs(got - want) < 1e-12, (got, want)


def test_opencv_corner_cases_by_hand():
    """cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + contourArea + boundingRect (connected_component_loss.py:28-56)
    on the shapes where pixel count and polygon area differ most.  Each case: expected (penalty sum, external contours)."""
    H, W = 48, 56
    ma, ed = 20, 6

    def run(m):
        d = _device(m[None], ed, ma)
        o = _oracle(m[None], ed, ma)
        assert abs(d[0] - o) < 1e-12, (d, o)
        return d

    # 1-pixel-wide horizontal line of 30 pixels: the contour runs out and back along the same pixels -> area 0
    m = np.zeros((H, W), bool); m[20, 10:40] = True
    assert run(m) == (1.0, 1.0)
    # 1-pixel-wide closed rectangular frame 21 x 31: the OUTER border polygon has area 20 * 30 = 600 (>= min_area), centre far
    # from the edges -> no penalty; its inside is a hole, not a contour
    m = np.zeros((H, W), bool); m[10, 10:41] = m[30, 10:41] = True; m[10:31, 10] = m[10:31, 40] = True
    assert run(m) == (0.0, 1.0)
    # two 5x5 squares touching only at a corner: ONE 8-connected component; outer border polygon = two 4x4 squares = 32
    m = np.zeros((H, W), bool); m[10:15, 10:15] = True; m[15:20, 15:20] = True
    assert run(m) == (0.0, 1.0)
    # ... and 3x3 squares the same way: area 4 + 4 = 8 < 20 -> 1 - 8/20
    m = np.zeros((H, W), bool); m[10:13, 10:13] = True; m[13:16, 13:16] = True
    p, c = run(m); assert c == 1.0 and abs(p - 0.6) < 1e-15
    # anti-diagonal staircase of single pixels: 8-connected, zero area
    m = np.zeros((H, W), bool)
    for i in range(9):
        m[30 - i, 12 + i] = True
    assert run(m) == (1.0, 1.0)
    # nested: ring (outer 25x25) > hole > island 9x9 ring > hole > 3x3 dot.  RETR_EXTERNAL reports the outermost ring only:
    # area 24 * 24 = 576
    m = np.zeros((H, W), bool)
    m[8:33, 8:33] = True; m[10:31, 10:31] = False
    m[16:25, 16:25] = True; m[18:23, 18:23] = False
    m[19:22, 19:22] = True
    assert run(m) == (0.0, 1.0)
    # a blob touching all four image borders (full frame of width 2): area (H-1)(W-1), bbox centre = image centre: d = min(28,
    # 28, 24, 24) = 24 >= 6 -> no penalty, one contour; the background island inside is a hole
    m = np.zeros((H, W), bool); m[:2] = m[-2:] = True; m[:, :2] = m[:, -2:] = True
    assert run(m) == (0.0, 1.0)
    # a large blob in the corner: area 15 * 11 = 165 >= 20, bbox (0,0,12,16): centre (6, 8) -> d = 6 is NOT < 6 -> no penalty;
    # one pixel narrower: centre x = 11 // 2 = 5 -> d = 5 -> 1 - 5/6
    m = np.zeros((H, W), bool); m[0:16, 0:12] = True
    assert run(m) == (0.0, 1.0)
    m = np.zeros((H, W), bool); m[0:16, 0:11] = True
    p, c = run(m); assert c == 1.0 and abs(p - (1 - 5 / 6)) < 1e-15
    # empty mask: no contours, zero loss; full mask: one contour of area (H-1)(W-1)
    assert run(np.zeros((H, W), bool)) == (0.0, 0.0)
    assert run(np.ones((H, W), bool)) == (0.0, 1.0)
