"""connected_component_loss -- drop-in surface of /root/reference/utils/connected_component_loss.py:7-60.

Like the reference, the mask is thresholded on the device (p > 0.5), copied to the host and analysed there; the
host analysis is `uh_cc_loss_host` in libunet_hip.so (Suzuki-Abe external contours, shoelace contourArea, integer
bounding box) instead of OpenCV.  Returns a Python float (no gradient), as the reference does.
PARITY UNPINNED: OpenCV is not available in this image; see tests/test_cc_loss.py for the
hand-derived known answers.
"""
from __future__ import annotations

import ctypes

import torch

from .._lib import LIB


def connected_component_loss(pred_mask, edge_distance=50, min_area=1000, penalty_weight=0.1):
    if pred_mask.dim() != 3:
        raise ValueError(f"pred_mask must be [B,H,W], got {tuple(pred_mask.shape)}")
    B, H, W = pred_mask.shape
    binary = (pred_mask.detach() > 0.5).to(torch.uint8).contiguous().cpu()      # connected_component_loss.py:25
    out = (ctypes.c_double * 2)()
    LIB.call("uh_cc_loss_host", binary.data_ptr(), B, H, W, int(edge_distance), int(min_area),
             ctypes.cast(out, ctypes.c_void_p))
    return float(out[0]) * penalty_weight                                        # :59
