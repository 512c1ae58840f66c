"""CPU restatement (scipy.ndimage) of /root/reference/utils/post_process.py -- TEST INFRASTRUCTURE ONLY (imported by
tests/ only; the product path is csrc/post_process.hip).  PARITY UNPINNED: the reference's arithmetic lives in OpenCV
(opencv-python~=4.11, not installed here); this restates the documented semantics of the calls it makes:

  remove_internal_regions (post_process.py:5-48): cv2.findContours(RETR_EXTERNAL) + cv2.drawContours(thickness=-1) fills
      everything inside the outer border of each 8-connected foreground component -> background pixels that cannot
      reach the image border through 4-connected background paths become foreground;
  cv2.morphologyEx(MORPH_OPEN, ones(k,k)) (:70-71): erosion (outside the image counts as foreground) then dilation;
  cv2.connectedComponentsWithStats(connectivity=8) (:74-80): components with area < min_area are dropped;
  (:84-86): kept pixels -> 2, every other pixel -> 0.
"""
import numpy as np
from scipy import ndimage


def remove_internal_regions(mask: np.ndarray) -> np.ndarray:
    fg = mask == 2
    lab, n = ndimage.label(~fg, structure=ndimage.generate_binary_structure(2, 1))      # 4-connected background
    border = np.zeros(n + 1, bool)
    for edge in (lab[0, :], lab[-1, :], lab[:, 0], lab[:, -1]):
        border[np.unique(edge)] = True
    border[0] = True                                                                     # label 0 = foreground
    out = mask.copy()
    out[~border[lab]] = 2
    return out


def postprocess_mask(mask: np.ndarray, min_area: int = 15000, morph_kernel_size: int = 3) -> np.ndarray:
    mask = remove_internal_regions(np.asarray(mask))
    fg = mask == 2
    k = np.ones((morph_kernel_size, morph_kernel_size), bool)
    opened = ndimage.binary_dilation(ndimage.binary_erosion(fg, structure=k, border_value=1), structure=k, border_value=0)
    lab, n = ndimage.label(opened, structure=np.ones((3, 3), bool))
    area = np.bincount(lab.ravel(), minlength=n + 1)
    keep = area >= min_area
    keep[0] = False
    out = np.zeros_like(mask)
    out[keep[lab]] = 2
    return out
