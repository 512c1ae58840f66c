"""connected_component_loss -- drop-in surface of /root/reference/utils/connected_component_loss.py:7-60.

The reference thresholds on the device (p > 0.5), copies every mask to the host and analyses it with OpenCV.  Here a GPU
tensor never leaves the device: `uh_cc_loss_device` fills holes and labels the 8-connected components by union-find, takes
cv2.contourArea as an integer sum over 2x2 pixel blocks and cv2.boundingRect by integer atomics, and returns the batch
penalty as one double (read back for the Python float the reference returns; no gradient).  A CPU tensor takes
`uh_cc_loss_host` (Suzuki-Abe border following + shoelace area on the host) -- the two agree exactly
(tests/test_gpu_cc_loss.py).
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
    if pred_mask.is_cuda:
        binary = (pred_mask.detach() > 0.5).to(torch.uint8).contiguous()         # connected_component_loss.py:25
        nbytes = LIB.query("uh_cc_loss_ws_bytes", B, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=binary.device)
        res = torch.empty(2, dtype=torch.float64, device=binary.device)
        LIB.call("uh_cc_loss_device", binary.data_ptr(), B, H, W, int(edge_distance), int(min_area), ws.data_ptr(), nbytes,
                 res.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return float(res[0].item()) * penalty_weight                             # :59
    binary = (pred_mask.detach() > 0.5).to(torch.uint8).contiguous().cpu()      # connected_component_loss.py:25
    out = (ctypes.c_double * 2)()
    LIB.call("uh_cc_loss_host", binary.data_ptr(), B, H, W, int(edge_distance), int(min_area),
             ctypes.cast(out, ctypes.c_void_p))
    return float(out[0]) * penalty_weight                                        # :59
