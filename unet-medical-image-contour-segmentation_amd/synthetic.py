"""Synthetic contour-segmentation batches: 1-3 random filled ellipses of class 2 on a class-1
background with a class-0 border band -- mask values {0,1,2} as data_loading.py:74-78 produces, images in [0,1]
(data_loading.py:86-87).  Gives a learnable, non-degenerate foreground after train.py:119's `true_masks //= 2`."""
from __future__ import annotations

import numpy as np
import torch


def ellipse_batch(batch: int, size: int, seed: int = 2, border: int = 4, noise: float = 0.15):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    images = np.zeros((batch, 1, size, size), np.float32)
    masks = np.ones((batch, size, size), np.int64)
    for b in range(batch):
        fg = np.zeros((size, size), bool)
        for _ in range(int(rng.integers(1, 4))):
            cy, cx = rng.uniform(0.25, 0.75, 2) * size
            ry, rx = rng.uniform(0.08, 0.22, 2) * size
            th = rng.uniform(0, np.pi)
            dy, dx = yy - cy, xx - cx
            u = (dx * np.cos(th) + dy * np.sin(th)) / rx
            v = (-dx * np.sin(th) + dy * np.cos(th)) / ry
            fg |= (u * u + v * v) <= 1.0
        m = masks[b]
        m[fg] = 2
        m[:border, :] = 0; m[-border:, :] = 0; m[:, :border] = 0; m[:, -border:] = 0
        img = np.where(m == 2, 0.75, np.where(m == 1, 0.30, 0.05)).astype(np.float32)
        img += rng.normal(0.0, noise, img.shape).astype(np.float32)
        images[b, 0] = np.clip(img, 0.0, 1.0)
    return torch.from_numpy(images), torch.from_numpy(masks)
