"""small modules of contrastyou/projectors/nn.py kept for state-dict / repr parity"""
from __future__ import annotations

from torch import nn
from torch.nn import functional as F


class Flatten(nn.Module):
    def forward(self, features):
        return features.view(features.shape[0], -1)


class Normalize(nn.Module):
    def __init__(self, dim=1) -> None:
        super().__init__()
        self._dim = dim

    def forward(self, input):  # noqa: A002
        return F.normalize(input, p=2, dim=self._dim)


class Identical(nn.Module):
    def forward(self, input):  # noqa: A002
        return input


class SoftmaxWithT(nn.Softmax):
    """softmax(x / T) (contrastyou/projectors/nn.py:35-44)"""

    def __init__(self, dim, T: float = 1.0) -> None:
        super().__init__(dim)
        self._T = T

    def forward(self, input):  # noqa: A002
        return super().forward(input / self._T)
