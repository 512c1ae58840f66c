"""Checkpoint wire format of the reference: a plain `torch.save`d state_dict with the
reference's key layout plus one extra entry 'mask_values' (train.py:208-216); loaders drop that entry before
`load_state_dict` (train.py:275-280, predict.py:106-109).  Weights trained by either side load in the other."""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch


def save_checkpoint(model: torch.nn.Module, path: str, mask_values: Optional[Iterable] = None) -> str:
    state_dict = {k: v.detach().to("cpu") for k, v in model.state_dict().items()}
    state_dict["mask_values"] = list(mask_values) if mask_values is not None else []
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    torch.save(state_dict, path)
    return path


def load_checkpoint(model: torch.nn.Module, path: str, device=None):
    """-> mask_values stored with the weights (or None).  Raises like load_state_dict on a key/shape mismatch."""
    try:
        state_dict = torch.load(path, map_location=device if device is not None else "cpu", weights_only=True)
    except Exception:
        state_dict = torch.load(path, map_location=device if device is not None else "cpu", weights_only=False)
    mask_values = state_dict.pop("mask_values", None)
    model.load_state_dict(state_dict)
    return mask_values
