"""`KL_div` / `Entropy` with the interface of contrastyou/losses/kl.py:31-135.

`KL_div.forward(prob, target)` keeps the reference's probability-space signature for arbitrary
callers.  The supervised loss of the epochers -- KL_div(softmax(logits), one_hot(labels)) with
mean reduction and no class weights (epocher.py:317-318) -- goes through
`KL_div.from_logits(logits, labels)`: one fused HIP pass (softmax + log + mean) forward and one
backward, with no one-hot tensor and no `unique()` host sync.
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch
from torch import Tensor, nn

from cyhip.functions import SoftmaxKLFn

__all__ = ["Entropy", "KL_div"]


def _check_reduction_params(reduction):
    assert reduction in ("mean", "sum", "none"), \
        f"reduction should be in ``'none'`` | ``'mean'`` | ``'sum'``. ``'none'``, given {reduction}"


class Entropy(nn.Module):
    """- sum_c p log(p + eps)"""

    def __init__(self, reduction="mean", eps=1e-16):
        super().__init__()
        _check_reduction_params(reduction)
        self._eps, self._reduction = eps, reduction

    def forward(self, input_: Tensor) -> Tensor:
        assert input_.dim() >= 2
        e = -(input_ * (input_ + self._eps).log()).sum(1)
        if self._reduction == "mean":
            return e.mean()
        if self._reduction == "sum":
            return e.sum()
        return e


class KL_div(nn.Module):
    """KL(target, prob) = - sum_c target * log((prob + eps) / (target + eps))"""

    def __init__(self, reduction="mean", eps=1e-16, weight: Union[List[float], Tensor] = None):
        super().__init__()
        _check_reduction_params(reduction)
        self._eps, self._reduction = eps, reduction
        self._weight: Optional[Tensor] = None
        if weight is not None:
            w = torch.as_tensor(weight).float()
            self._weight = w / w.sum() * len(w)

    @property
    def fusable(self) -> bool:
        return self._reduction == "mean" and self._weight is None

    def from_logits(self, logits: Tensor, labels: Tensor) -> Tensor:
        """== self(logits.softmax(1), one_hot(labels)) for the fusable configuration"""
        if not self.fusable:
            C = logits.shape[1]
            onehot = torch.nn.functional.one_hot(labels.long(), C).movedim(-1, 1)
            return self(logits.softmax(1), onehot)
        return SoftmaxKLFn.apply(logits, labels, float(self._eps))

    def forward(self, prob: Tensor, target: Tensor, **kwargs) -> Tensor:
        b, c, *hwd = target.shape
        kl = -target * torch.log((prob + self._eps) / (target + self._eps))
        if self._weight is not None:
            assert len(self._weight) == c
            shape = [1, c] + [1] * len(hwd)
            kl = kl * self._weight.to(kl.device).view(*shape)
        kl = kl.sum(1)
        if self._reduction == "mean":
            return kl.mean()
        if self._reduction == "sum":
            return kl.sum()
        return kl

    def __repr__(self):
        return f"{self.__class__.__name__}\n, weight={self._weight}"
