"""small shared type helpers (subset of contrastyou/types.py of the reference)"""
from __future__ import annotations

from typing import Any

import torch


def to_device(obj: Any, device, non_blocking: bool = True) -> Any:
    """recursively move tensors of a nested list/tuple/dict to `device` (types.py:366)"""
    if isinstance(obj, torch.Tensor):
        return obj.to(device, non_blocking=non_blocking)
    if isinstance(obj, dict):
        return {k: to_device(v, device, non_blocking) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_device(v, device, non_blocking) for v in obj)
    return obj


def to_float(value) -> float:
    if isinstance(value, torch.Tensor):
        return float(value.item())
    return float(value)
