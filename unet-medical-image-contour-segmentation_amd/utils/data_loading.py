"""Directory dataset with the reference's item contract (behaviour of
/root/reference/utils/data_loading.py:14-136, pinned bit for bit by fixture G12):

    ds = BasicDataset(images_dir, mask_dir, scale=1.0, mask_suffix='_mask', augment=True)
    len(ds)        = files x 4 quarter-turn rotations when augment (item i: file i // 4, rotation i % 4), else files
    ds[i]          = {'image': float32 [C, H, W] in [0, 1], 'mask': int64 [H, W] in {0, 1, 2}}
    ds.mask_values = sorted raw grey levels found in the mask files (travels inside checkpoints, train.py:213)

Masks are grey images coded 0 = ghost, 128 = background, 255 = contour -> classes 0 / 1 / 2 through a 256-entry lookup
table; images are rescaled with BICUBIC, masks with NEAREST, and an image is divided by 255 only when it holds a value
above 1.  Host-side (PIL + numpy): feeding the GPU path is the DataLoader's job.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, List

import numpy as np
import torch
from torch.utils.data import Dataset

QUARTER_TURNS = 4
_CLASS_OF_GREY = np.zeros(256, dtype=np.int8)
_CLASS_OF_GREY[128] = 1
_CLASS_OF_GREY[255] = 2


def load_image(filename):
    """PIL image from an image file, a .npy array or a saved tensor (.pt / .pth)."""
    from PIL import Image
    suffix = Path(str(filename)).suffix
    if suffix == ".npy":
        return Image.fromarray(np.load(filename))
    if suffix in {".pt", ".pth"}:
        return Image.fromarray(torch.load(filename).numpy())
    return Image.open(filename)


def _grey_levels(path) -> np.ndarray:
    arr = np.asarray(load_image(path))
    if arr.ndim not in (2, 3):
        raise ValueError(f"Loaded masks should have 2 or 3 dimensions, found {arr.ndim}")
    return np.unique(arr) if arr.ndim == 2 else np.unique(arr.reshape(-1, arr.shape[-1]), axis=0)


def _rescaled(pil_img, scale: float, resample):
    width, height = pil_img.size
    size = (int(scale * width), int(scale * height))
    assert min(size) > 0, "Scale is too small, resized images would have no pixel"
    return pil_img.resize(size, resample=resample)


def _quarter_turn(pil_img, turns: int):
    """`turns` x 90 degrees counter-clockwise with the canvas following the image (what Image.rotate(angle, expand=True)
    does for right angles: a lossless transpose)."""
    from PIL import Image
    op = {1: Image.ROTATE_90, 2: Image.ROTATE_180, 3: Image.ROTATE_270}.get(turns % QUARTER_TURNS)
    return pil_img if op is None else pil_img.transpose(op)


class BasicDataset(Dataset):
    def __init__(self, images_dir: str, mask_dir: str, scale: float = 1.0, mask_suffix: str = "_mask", augment=True):
        assert 0 < scale <= 1, "Scale must be between 0 and 1"
        self.images_dir, self.mask_dir = Path(images_dir), Path(mask_dir)
        self.scale, self.mask_suffix, self.augment = scale, mask_suffix, augment
        self.ids: List[str] = [os.path.splitext(name)[0] for name in os.listdir(images_dir)
                               if not name.startswith(".") and (self.images_dir / name).is_file()]
        if not self.ids:
            raise RuntimeError(f"No input file found in {images_dir}, make sure you put your images there")
        levels = [_grey_levels(self._only(self.mask_dir, stem + mask_suffix, "mask")) for stem in self.ids]
        self.mask_values = sorted(np.unique(np.concatenate(levels), axis=0).tolist())

    @staticmethod
    def _only(folder: Path, stem: str, what: str) -> Path:
        hits = list(folder.glob(stem + ".*"))
        assert len(hits) == 1, f"Either no {what} or multiple {what}s found for the ID {stem}: {hits}"
        return hits[0]

    def __len__(self) -> int:
        return len(self.ids) * (QUARTER_TURNS if self.augment else 1)

    @staticmethod
    def preprocess(mask_values, pil_img, scale, is_mask):
        """Same signature as the reference's static method (predict.py:19 calls it with mask_values=None)."""
        from PIL import Image
        if is_mask:
            grey = np.asarray(_rescaled(pil_img, scale, Image.NEAREST))
            return _CLASS_OF_GREY[grey]
        arr = np.asarray(_rescaled(pil_img, scale, Image.BICUBIC))
        chw = arr[None] if arr.ndim == 2 else np.moveaxis(arr, -1, 0)
        return chw.astype(np.float32) / 255.0 if (chw > 1).any() else chw

    def __getitem__(self, index: int) -> Dict[str, torch.Tensor]:
        views = QUARTER_TURNS if self.augment else 1
        stem, turns = self.ids[index // views], index % views
        image = load_image(self._only(self.images_dir, stem, "image"))
        mask = load_image(self._only(self.mask_dir, stem + self.mask_suffix, "mask"))
        assert image.size == mask.size, f"Image and mask {stem} should be the same size, but are {image.size} and {mask.size}"
        image, mask = _quarter_turn(image, turns), _quarter_turn(mask, turns)
        pixels = self.preprocess(self.mask_values, image, self.scale, is_mask=False)
        classes = self.preprocess(self.mask_values, mask, self.scale, is_mask=True)
        return {"image": torch.from_numpy(np.ascontiguousarray(pixels)).float(),
                "mask": torch.from_numpy(np.ascontiguousarray(classes)).long()}


class CarvanaDataset(BasicDataset):
    def __init__(self, images_dir, mask_dir, scale=1, augment=True):
        super().__init__(images_dir, mask_dir, scale, mask_suffix="_mask", augment=augment)
