"""CPU restatement (numpy) of what /root/reference/utils/data_loading.py does to a DECODED image / mask pair -- TEST
INFRASTRUCTURE ONLY (imported by tests/ and bench.py's loader comparison; the product path is csrc/data_prep.hip).
Pinned by fixture G12 (tests/test_data_prep_cpu.py): the reference's own BasicDataset items at scale 1.0.

  data_loading.py:100-121  Image.rotate(angle, expand=True) for angle = 90 * turns: a quarter-turn transpose,
                           counter-clockwise, i.e. np.rot90(a, turns) on the first two axes;
  :66-70                   resize to the same size (scale 1) is the identity;
  :74-78                   mask: 255 -> 2, 128 -> 1, anything else -> 0 (int8 there, .long() at :131);
  :80-87                   image: [H,W] -> [1,H,W], [H,W,C] -> [C,H,W]; divided by 255.0 as float32 only when a value > 1 exists;
  :129-132                 float32 image, int64 mask.
"""
import numpy as np


def prepare_item(image_u8: np.ndarray, mask_u8: np.ndarray, turns: int = 0):
    """-> (float32 [C,H',W'], int64 [H',W']) for one decoded pair."""
    img = np.rot90(np.asarray(image_u8), turns & 3, axes=(0, 1))
    grey = np.rot90(np.asarray(mask_u8), turns & 3, axes=(0, 1))
    mask = np.zeros(grey.shape, dtype=np.int8)
    mask[grey == 255] = 2
    mask[grey == 128] = 1
    img = img[np.newaxis, ...] if img.ndim == 2 else img.transpose((2, 0, 1))
    if (img > 1).any():
        img = img.astype(np.float32) / 255.0
    return np.ascontiguousarray(img, dtype=np.float32), np.ascontiguousarray(mask).astype(np.int64)


def prepare_batch(images_u8, masks_u8, turns=None):
    """Batch form: lists / arrays of decoded pairs -> (float32 [B,C,H',W'], int64 [B,H',W'])."""
    B = len(images_u8)
    turns = [0] * B if turns is None else list(turns)
    items = [prepare_item(images_u8[b], masks_u8[b], int(turns[b])) for b in range(B)]
    return np.stack([i for i, _ in items]), np.stack([m for _, m in items])
