"""Oracle (test infrastructure): numpy/scipy restatement of connected_component_loss
(/root/reference/utils/connected_component_loss.py:7-60).

PARITY UNPINNED: the arithmetic of the reference lives in OpenCV (opencv-python~=4.11.0.86, requirements.txt:8),
which is neither vendored in /root/reference nor installed here.  This file restates the published algorithms
independently of the C++ in csrc/cc_loss.hip: components by scipy.ndimage.label (8-connectivity), "external" =
touching the frame-connected background (4-connectivity), outer border by Moore-neighbour tracing, contourArea by the
shoelace formula over the pixel-centre chain, boundingRect = component bounding box.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

# clockwise from West (image coordinates, y down): W, NW, N, NE, E, SE, S, SW
_NB = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1)]


def _trace_area(comp: np.ndarray) -> float:
    """|shoelace| of the outer border chain of a single 8-connected component (bool array, zero padded)."""
    ys, xs = np.nonzero(comp)
    order = np.lexsort((xs, ys))
    sx, sy = int(xs[order[0]]), int(ys[order[0]])          # first pixel in raster order: on the outer border
    if comp.sum() == 1:
        return 0.0
    # Moore tracing, backtrack starts at the West neighbour (background by construction)
    cx, cy, back = sx, sy, 0
    pts = [(cx, cy)]
    first_move = None
    for _ in range(8 * comp.size + 16):
        found = None
        for step in range(1, 9):                            # scan clockwise starting after the backtrack direction
            k = (back + step) % 8
            nx, ny = cx + _NB[k][0], cy + _NB[k][1]
            if comp[ny, nx]:
                found = (nx, ny, k)
                break
        nx, ny, k = found
        move = (cx, cy, nx, ny)
        if first_move is None:
            first_move = move
        elif move == first_move:
            break
        pts.append((nx, ny))
        # Moore: restart the clockwise scan just after the direction that points back at the pixel we came from
        back = (k + 5) % 8
        cx, cy = nx, ny
    a = 0.0
    for i in range(len(pts) - 1):
        a += pts[i][0] * pts[i + 1][1] - pts[i + 1][0] * pts[i][1]
    return abs(a) * 0.5


def external_components(mask: np.ndarray):
    m = np.pad(mask.astype(bool), 1)
    lab, n = ndimage.label(m, structure=np.ones((3, 3)))
    bg, _ = ndimage.label(~m)                                # 4-connectivity
    outside = bg == bg[0, 0]
    touch = ndimage.binary_dilation(outside, structure=ndimage.generate_binary_structure(2, 1))
    out = []
    for k in range(1, n + 1):
        comp = lab == k
        if not (comp & touch).any():
            continue                                         # enclosed by another component: not RETR_EXTERNAL
        ys, xs = np.nonzero(comp)
        out.append({"area": _trace_area(comp), "x": int(xs.min()) - 1, "y": int(ys.min()) - 1,
                    "w": int(xs.max() - xs.min() + 1), "h": int(ys.max() - ys.min() + 1)})
    return out


def connected_component_loss(pred: np.ndarray, edge_distance=50, min_area=1000, penalty_weight=0.1) -> float:
    B, H, W = pred.shape
    total = 0.0
    for b in range(B):
        for c in external_components(pred[b] > 0.5):
            if c["area"] < min_area:
                total += 1.0 - c["area"] / min_area
                continue
            cx, cy = c["x"] + c["w"] // 2, c["y"] + c["h"] // 2
            d = min(cx, W - cx, cy, H - cy)
            if d < edge_distance:
                total += 1.0 - d / edge_distance
    return total / B * penalty_weight
