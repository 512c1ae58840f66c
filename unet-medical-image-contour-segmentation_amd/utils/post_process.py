"""Mask post-processing of /root/reference/utils/post_process.py on the device (csrc/post_process.hip):

    postprocess_mask(mask, min_area=15000, morph_kernel_size=3)      post_process.py:51-88
    remove_internal_regions(mask, foreground_value=2)                post_process.py:5-48

`mask` holds class indices {0,1,2}: a torch tensor on the GPU ([H,W] or [B,H,W], any integer dtype; returned in the
same dtype/shape) or a numpy array (the reference's calling convention; copied to the GPU and back).  The reference
runs OpenCV on the host image by image; here the whole batch is processed by one call.  PARITY UNPINNED: OpenCV is not
installed; the tests compare with a scipy restatement of the same semantics."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .._lib import LIB


def _run(mask_u8: torch.Tensor, min_area: int, ksize: int) -> torch.Tensor:
    B, H, W = mask_u8.shape
    out = torch.empty_like(mask_u8)
    nbytes = LIB.query("uh_postprocess_ws_bytes", B, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=mask_u8.device)
    LIB.call("uh_postprocess_masks", mask_u8.data_ptr(), out.data_ptr(), B, H, W, int(min_area), int(ksize),
             ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    return out


def _to_device_u8(mask):
    is_np = isinstance(mask, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(mask)) if is_np else mask
    if is_np:
        if not torch.cuda.is_available():
            raise RuntimeError("postprocess_mask needs an MI355X: the HIP path has no CPU fallback")
        t = t.cuda()
    ops._require_gpu(t, "mask")
    squeeze = t.dim() == 2
    if squeeze:
        t = t.unsqueeze(0)
    if t.dim() != 3:
        raise RuntimeError(f"mask must be [H,W] or [B,H,W], got {tuple(t.shape)}")
    return t.to(torch.uint8).contiguous(), is_np, squeeze


def _back(out, like, is_np, squeeze):
    if squeeze:
        out = out.squeeze(0)
    if is_np:
        return out.cpu().numpy().astype(like.dtype)
    return out.to(like.dtype)


def postprocess_mask(mask, min_area: int = 15000, morph_kernel_size: int = 3):
    t, is_np, squeeze = _to_device_u8(mask)
    return _back(_run(t, min_area, morph_kernel_size), mask, is_np, squeeze)


def remove_internal_regions(mask, foreground_value: int = 2, background_values=(0, 1)):
    if foreground_value != 2:
        raise NotImplementedError("the device kernel fills the class-2 foreground (the only use in the reference)")
    t, is_np, squeeze = _to_device_u8(mask)
    filled = _run(t, 1, 1) == 2                      # 1x1 opening, every component kept: the hole-filled foreground
    out = torch.where(filled, torch.full_like(t, 2), t)
    return _back(out, mask, is_np, squeeze)
