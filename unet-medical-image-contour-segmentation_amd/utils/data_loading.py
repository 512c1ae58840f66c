"""Directory dataset with the reference's item contract (behaviour of
/root/reference/utils/data_loading.py:14-136, pinned bit for bit by fixture G12):

    ds = BasicDataset(images_dir, mask_dir, scale=1.0, mask_suffix='_mask', augment=True)
    len(ds)        = files x 4 quarter-turn rotations when augment (item i: file i // 4, rotation i % 4), else files
    ds[i]          = {'image': float32 [C, H, W] in [0, 1], 'mask': int64 [H, W] in {0, 1, 2}}
    ds.mask_values = sorted raw grey levels found in the mask files (travels inside checkpoints, train.py:213)

Masks are grey images coded 0 = ghost, 128 = background, 255 = contour -> classes 0 / 1 / 2 through a 256-entry lookup
table; images are rescaled with BICUBIC, masks with NEAREST, and an image is divided by 255 only when it holds a value
above 1.  `ds[i]` is the reference's host path (PIL + numpy).

The device stage (csrc/data_prep.hip, `uh_batch_prepare`): `ds.raw_item(i)` stops after the decode (and, for scale < 1,
the host-side rotation + rescale) and returns uint8 pixels + the number of quarter turns still to apply;
`collate_raw` stacks such items into pinned uint8 batches and `prepare_batch_device` turns a batch into the NHWC fp32 /
bf16 image tensor in [0, 1] and the int64 labels ON THE GPU -- rotation by index, the per-image /255 rule and the label
remap in one kernel -- so a batch crosses PCIe at 1 + C bytes per pixel instead of 4C + 8 and no PIL / numpy arithmetic
runs per pixel on the host (above ~1 000 img/s the PIL loader is the bottleneck).
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


    def raw_item(self, index: int) -> Dict[str, object]:
        """Item `index` up to the point where arithmetic starts: uint8 pixels [H, W] or [H, W, C], uint8 mask grey levels
        [H, W] and `turns`, the quarter turns the device stage still has to apply.  At scale 1 the files are decoded and
        nothing else (the rescale to the same size is the identity, data_loading.py:66-70); at scale < 1 the host rotates
        and rescales exactly like `__getitem__` (PIL's resampling is not restated on the device) and turns = 0."""
        from PIL import Image
        views = QUARTER_TURNS if self.augment else 1
        stem, turns = self.ids[index // views], index % views
        image = load_image(self._only(self.images_dir, stem, "image"))
        mask = load_image(self._only(self.mask_dir, stem + self.mask_suffix, "mask"))
        assert image.size == mask.size, f"Image and mask {stem} should be the same size, but are {image.size} and {mask.size}"
        if self.scale != 1:
            image = _rescaled(_quarter_turn(image, turns), self.scale, Image.BICUBIC)
            mask = _rescaled(_quarter_turn(mask, turns), self.scale, Image.NEAREST)
            turns = 0
        pixels, grey = np.asarray(image), np.asarray(mask)
        if pixels.dtype != np.uint8 or grey.dtype != np.uint8:
            raise TypeError(f"raw_item: {stem} does not decode to 8-bit pixels ({pixels.dtype}, {grey.dtype}); use ds[i]")
        return {"image_u8": pixels, "mask_u8": grey, "turns": int(turns)}


def collate_raw(items, pin: bool = True) -> Dict[str, torch.Tensor]:
    """Stack `raw_item`s into uint8 batches [B,H,W,C] / [B,H,W] (pinned host memory) + an int32 turn table.  Every item
    must produce the same output shape: equal decoded sizes and, unless the images are square, turn counts of one parity."""
    imgs = [it["image_u8"] if it["image_u8"].ndim == 3 else it["image_u8"][..., None] for it in items]
    shape = imgs[0].shape
    if any(a.shape != shape for a in imgs) or any(it["mask_u8"].shape != shape[:2] for it in items):
        raise ValueError("collate_raw: the items of a batch must have one decoded size")
    turns = [int(it["turns"]) & 3 for it in items]
    if shape[0] != shape[1] and len({t & 1 for t in turns}) > 1:
        raise ValueError("collate_raw: non-square images rotated by odd and even quarter turns do not stack")
    B = len(items)
    image = torch.empty((B,) + shape, dtype=torch.uint8, pin_memory=pin and torch.cuda.is_available())
    mask = torch.empty((B,) + shape[:2], dtype=torch.uint8, pin_memory=pin and torch.cuda.is_available())
    for b, (a, it) in enumerate(zip(imgs, items)):
        np.copyto(image[b].numpy(), a)
        np.copyto(mask[b].numpy(), it["mask_u8"])
    return {"image_u8": image, "mask_u8": mask, "turns": torch.tensor(turns, dtype=torch.int32)}


def prepare_batch_device(image_u8: torch.Tensor, mask_u8: torch.Tensor, turns=None, device=None,
                         dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
    """uint8 batch (host, ideally pinned, or already on the GPU) -> {'image': logical [B,C,H,W] tensor in `dtype`
    (channels_last memory, values as data_loading.py:86-87 produces them), 'mask': int64 [B,H,W] classes}: what
    train.py:113-114 hands to the model, with rotation / normalisation / label remap done by ONE kernel on the GPU
    (`uh_batch_prepare`).  No CPU fallback."""
    from .. import ops
    from .._lib import LIB
    dev = torch.device(device) if device is not None else (image_u8.device if image_u8.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    if dev.type != "cuda":
        raise RuntimeError("prepare_batch_device needs a GPU (the host path is BasicDataset.__getitem__)")
    if image_u8.dtype != torch.uint8 or mask_u8.dtype != torch.uint8:
        raise TypeError("prepare_batch_device takes uint8 batches (BasicDataset.raw_item / collate_raw)")
    if image_u8.dim() == 3:
        image_u8 = image_u8.unsqueeze(-1)
    B, H, W, C = image_u8.shape
    if tuple(mask_u8.shape) != (B, H, W) or not 1 <= C <= 4:
        raise ValueError(f"prepare_batch_device: image batch {tuple(image_u8.shape)} / mask batch {tuple(mask_u8.shape)}")
    tl = None
    odd = 0
    if turns is not None:
        tl = [int(t) & 3 for t in (turns.tolist() if torch.is_tensor(turns) else turns)]
        if len(tl) != B:
            raise ValueError("prepare_batch_device: one turn count per item")
        par = {t & 1 for t in tl}
        if H != W and len(par) > 1:
            raise ValueError("prepare_batch_device: non-square images rotated by odd and even quarter turns do not stack")
        odd = 1 if (H != W and par == {1}) else 0
        if not any(tl):
            tl = None
    with torch.cuda.device(dev):
        img_d = image_u8.contiguous().to(dev, non_blocking=True)
        msk_d = mask_u8.contiguous().to(dev, non_blocking=True)
        t_d = torch.tensor(tl, dtype=torch.int32).to(dev, non_blocking=True) if tl is not None else None
        Ho, Wo = (W, H) if odd else (H, W)
        image = torch.empty((B, Ho, Wo, C), dtype=dtype, device=dev)
        labels = torch.empty((B, Ho, Wo), dtype=torch.int64, device=dev)
        flags = torch.empty(B, dtype=torch.int32, device=dev)
        LIB.call("uh_batch_prepare", img_d.data_ptr(), C, msk_d.data_ptr(), None if t_d is None else t_d.data_ptr(), odd,
                 image.data_ptr(), C, labels.data_ptr(), flags.data_ptr(), B, H, W, ops._dt(image),
                 torch.cuda.current_stream().cuda_stream)
    return {"image": image.permute(0, 3, 1, 2), "mask": labels}


class CarvanaDataset(BasicDataset):
    def __init__(self, images_dir, mask_dir, scale=1, augment=True):
        super().__init__(images_dir, mask_dir, scale, mask_suffix="_mask", augment=augment)
