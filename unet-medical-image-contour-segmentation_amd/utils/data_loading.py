"""Directory dataset of the reference (/root/reference/utils/data_loading.py:14-136): grey-level
images + masks coded {0: ghost, 128: background, 255: contour}, optional x4 rotation augmentation, BICUBIC / NEAREST
rescale, label remap to {0, 1, 2}.  Host-side (PIL + numpy), same constructor / item contract:

    BasicDataset(images_dir, mask_dir, scale=1.0, mask_suffix='_mask', augment=True)
    len(ds) == n_files * (4 if augment else 1);  ds[i] -> {'image': float32 [C,H,W], 'mask': int64 [H,W]}
    ds.mask_values   sorted unique raw mask values over the directory (stored in checkpoints, train.py:213)
"""
from __future__ import annotations

from os import listdir
from os.path import isfile, join, splitext
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

ROTATIONS = (0, 90, 180, 270)          # data_loading.py:108-111: index // 4 = file, index % 4 = rotation


def load_image(filename):
    from PIL import Image
    ext = splitext(str(filename))[1]
    if ext == ".npy":
        return Image.fromarray(np.load(filename))
    if ext in (".pt", ".pth"):
        return Image.fromarray(torch.load(filename).numpy())
    return Image.open(filename)


def _unique_values(mask_file) -> np.ndarray:
    m = np.asarray(load_image(mask_file))
    if m.ndim == 2:
        return np.unique(m)
    if m.ndim == 3:
        return np.unique(m.reshape(-1, m.shape[-1]), axis=0)
    raise ValueError(f"Loaded masks should have 2 or 3 dimensions, found {m.ndim}")


class BasicDataset(Dataset):
    def __init__(self, images_dir: str, mask_dir: str, scale: float = 1.0, mask_suffix: str = "_mask", augment=True):
        self.images_dir = Path(images_dir)
        self.mask_dir = Path(mask_dir)
        assert 0 < scale <= 1, "Scale must be between 0 and 1"
        self.scale = scale
        self.mask_suffix = mask_suffix
        self.augment = augment
        self.ids = [splitext(f)[0] for f in listdir(images_dir) if isfile(join(images_dir, f)) and not f.startswith(".")]
        if not self.ids:
            raise RuntimeError(f"No input file found in {images_dir}, make sure you put your images there")
        uniq = [_unique_values(list(self.mask_dir.glob(i + self.mask_suffix + ".*"))[0]) for i in self.ids]
        self.mask_values = list(sorted(np.unique(np.concatenate(uniq), axis=0).tolist()))

    def __len__(self):
        return len(self.ids) * (len(ROTATIONS) if self.augment else 1)

    @staticmethod
    def preprocess(mask_values, pil_img, scale, is_mask):
        from PIL import Image
        w, h = pil_img.size
        new_w, new_h = int(scale * w), int(scale * h)
        assert new_w > 0 and new_h > 0, "Scale is too small, resized images would have no pixel"
        pil_img = pil_img.resize((new_w, new_h), resample=Image.NEAREST if is_mask else Image.BICUBIC)
        img = np.asarray(pil_img)
        if is_mask:
            mask = np.zeros((new_h, new_w), dtype=np.int8)
            mask[img == 255] = 2
            mask[img == 128] = 1
            return mask
        img = img[np.newaxis, ...] if img.ndim == 2 else img.transpose((2, 0, 1))
        if (img > 1).any():
            img = img.astype(np.float32) / 255.0
        return img

    def __getitem__(self, idx):
        n_rot = len(ROTATIONS) if self.augment else 1
        name = self.ids[idx // n_rot]
        angle = ROTATIONS[idx % n_rot]
        mask_file = list(self.mask_dir.glob(name + self.mask_suffix + ".*"))
        img_file = list(self.images_dir.glob(name + ".*"))
        assert len(img_file) == 1, f"Either no image or multiple images found for the ID {name}: {img_file}"
        assert len(mask_file) == 1, f"Either no mask or multiple masks found for the ID {name}: {mask_file}"
        mask = load_image(mask_file[0])
        img = load_image(img_file[0])
        assert img.size == mask.size, f"Image and mask {name} should be the same size, but are {img.size} and {mask.size}"
        if angle:
            img, mask = img.rotate(angle, expand=True), mask.rotate(angle, expand=True)
        img = self.preprocess(self.mask_values, img, self.scale, is_mask=False)
        mask = self.preprocess(self.mask_values, mask, self.scale, is_mask=True)
        assert np.isin(mask, (0, 1, 2)).all(), "mask holds an illegal class index"
        return {"image": torch.as_tensor(img.copy()).float().contiguous(),
                "mask": torch.as_tensor(mask.copy()).long().contiguous()}


class CarvanaDataset(BasicDataset):
    def __init__(self, images_dir, mask_dir, scale=1, augment=True):
        super().__init__(images_dir, mask_dir, scale, mask_suffix="_mask", augment=augment)
