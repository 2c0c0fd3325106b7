"""Running averages (contrastyou/meters/averagemeter.py).  Values may be 0-dim DEVICE tensors:
they are accumulated on the device and only read back in summary(), so adding a loss does not
force a host sync per batch (the reference calls `.item()` at every add)."""
from __future__ import annotations

import typing as t
from collections import defaultdict

import numpy as np
import torch

from .metric import Metric


class AverageValueMeter(Metric):
    def __init__(self):
        super().__init__()
        self.reset()

    def reset(self):
        self.sum = 0
        self.n = 0

    def _add(self, value, n=1):
        if isinstance(value, torch.Tensor):
            value = value.detach()
            if value.dim():
                value = value.reshape(-1)[0]
            value = value.double() if value.is_floating_point() else value
        self.sum = self.sum + value * n
        self.n += n

    def _summary(self):
        if self.n == 0:
            return np.nan
        s = self.sum
        return float(s.item() / self.n) if isinstance(s, torch.Tensor) else float(s / self.n)


class AverageValueDictionaryMeter(Metric):
    def __init__(self) -> None:
        super().__init__()
        self._meter_dicts: t.Dict[str, AverageValueMeter] = defaultdict(AverageValueMeter)

    def reset(self):
        for v in self._meter_dicts.values():
            v.reset()

    def _add(self, **kwargs):
        for k, v in kwargs.items():
            self._meter_dicts[k].add(v)

    def _summary(self):
        return {k: v.summary() for k, v in self._meter_dicts.items()}


class AverageValueListMeter(AverageValueDictionaryMeter):
    def _add(self, list_value: t.Iterable[float] = None, **kwargs):
        assert isinstance(list_value, t.Iterable)
        for i, v in enumerate(list_value):
            self._meter_dicts[str(i)].add(v)
