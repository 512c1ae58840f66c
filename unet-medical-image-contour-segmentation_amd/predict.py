"""Single-image inference of /root/reference/predict.py on the HIP path.

    predict_img(model, full_img, device) -> np.ndarray[H, W] of class indices        predict.py:15-29
    mask_to_image(mask) -> PIL.Image (0 / 128 / 255 grey levels)                      predict.py:52-58
    preprocess_image(pil_img, scale=1.0) -> float32 [C, H, W]                         data_loading.py:65-91 (image branch)

The forward runs the eval-mode kernels (BatchNorm running statistics folded into per-channel scale/shift),
argmax is `uh_argmax_classes`.  The CLI / file handling / matplotlib parts of predict.py are outside the scope.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import ops


def preprocess_image(pil_img, scale: float = 1.0) -> np.ndarray:
    """Image branch of BasicDataset.preprocess (data_loading.py:65-91); ndarrays are taken as already-decoded images."""
    from .utils.data_loading import BasicDataset
    if isinstance(pil_img, np.ndarray):
        if scale != 1.0:
            raise ValueError("scale != 1 needs a PIL image")
        img = pil_img[np.newaxis, ...] if pil_img.ndim == 2 else pil_img.transpose((2, 0, 1))
        return img.astype(np.float32) / 255.0 if (img > 1).any() else img
    return BasicDataset.preprocess(None, pil_img, scale, is_mask=False)


def predict_img(model, full_img, device):
    model.eval()
    img = torch.from_numpy(np.ascontiguousarray(preprocess_image(full_img, 1.0)))
    img = img.unsqueeze(0).to(device=device, dtype=torch.float32, memory_format=torch.channels_last)
    if hasattr(full_img, "size") and not isinstance(full_img, np.ndarray):
        out_hw = (full_img.size[1], full_img.size[0])
    else:
        out_hw = tuple(np.asarray(full_img).shape[:2])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=True):
        mask_pred = model(img)
        if tuple(mask_pred.shape[-2:]) != tuple(out_hw):
            # predict.py:26; never taken with scale=1 (the UNet output has the input's size, unet_parts.py:85-88)
            mask_pred = F.interpolate(mask_pred.float(), out_hw, mode="bilinear")
        idx = ops.argmax_classes(mask_pred).squeeze(0)
    return idx.cpu().numpy()


def mask_to_image(mask: np.ndarray):
    from PIL import Image
    vis = np.zeros_like(mask, dtype=np.uint8)
    vis[mask == 1] = 128
    vis[mask == 2] = 255
    return Image.fromarray(vis)
