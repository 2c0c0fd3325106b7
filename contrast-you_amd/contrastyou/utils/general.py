"""tensor predicates / conversions (subset of contrastyou/utils/general.py)"""
from __future__ import annotations

from typing import Iterable, Set

import torch
import torch.nn.functional as F
from torch import Tensor

__all__ = ["uniq", "sset", "simplex", "one_hot", "class2one_hot", "probs2one_hot", "probs2class"]


def uniq(a: Tensor) -> Set:
    return set(a.unique().tolist())


def sset(a: Tensor, sub: Iterable) -> bool:
    return uniq(a).issubset(set(sub))


def simplex(t: Tensor, axis=1) -> bool:
    s = t.sum(axis).float()
    return torch.allclose(s, torch.ones_like(s), rtol=1e-4, atol=1e-4)


def one_hot(t: Tensor, axis=1) -> bool:
    return simplex(t, axis) and sset(t, [0, 1])


def class2one_hot(seg: Tensor, C: int, class_dim: int = 1, check: bool = False) -> Tensor:
    """F.one_hot moved to `class_dim` (general.py:114-120).  The reference's label-set assert is a
    host sync (`unique()`); it is opt-in here (`check=True`)."""
    if check:
        assert sset(seg, list(range(C)))
    return F.one_hot(seg.long(), C).movedim(-1, class_dim)


def probs2class(probs: Tensor, class_dim: int = 1) -> Tensor:
    return probs.argmax(dim=class_dim)


def probs2one_hot(probs: Tensor, class_dim: int = 1) -> Tensor:
    C = probs.shape[class_dim]
    return class2one_hot(probs2class(probs, class_dim), C, class_dim)
