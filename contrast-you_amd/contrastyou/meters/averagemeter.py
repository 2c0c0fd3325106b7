"""Running means (the meters of contrastyou/meters/averagemeter.py).

Difference from the reference: a value may be a 0-dim DEVICE tensor.  It is then accumulated on the
device (f64) and read back only when `summary()` is called, so metering a loss does not force a host
synchronisation per batch (the reference calls `.item()` on every add).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable

import torch

from .metric import Metric


def _as_accumulable(v):
    """python number stays; a tensor becomes a detached 0-dim f64 (or integer) tensor"""
    if not isinstance(v, torch.Tensor):
        return v
    v = v.detach()
    if v.dim() > 0:
        v = v.reshape(-1)[0]
    return v.double() if v.is_floating_point() else v


class AverageValueMeter(Metric):
    """weighted running mean of scalars.  Device scalars are queued (no launch per add: `v.double() * n` and the
    running `sum + ...` were three tiny kernels per meter and step) and summed in f64 when the summary is asked
    for, or every 1024 values."""

    _FLUSH = 1024

    def __init__(self):
        super().__init__()
        self.sum, self.n = 0, 0
        self._queue = []

    def reset(self):
        self.sum, self.n = 0, 0
        self._queue = []

    def _add(self, value, n=1):
        if isinstance(value, torch.Tensor) and value.is_floating_point():
            v = value.detach()
            self._queue.append((v.reshape(-1)[0] if v.dim() > 0 else v, n))
            if len(self._queue) >= self._FLUSH:
                self._flush()
        else:
            self.sum = self.sum + _as_accumulable(value) * n
        self.n += n

    def _flush(self):
        q, self._queue = self._queue, []
        if not q:
            return
        by_dev = {}
        for v, n in q:
            by_dev.setdefault((v.device, v.dtype), []).append((v, n))
        for (dev, _), items in by_dev.items():
            vals = torch.stack([v for v, _ in items]).double()
            if any(n != 1 for _, n in items):
                vals = vals * torch.tensor([float(n) for _, n in items], dtype=torch.float64).to(dev)
            total = vals.sum()
            if isinstance(self.sum, torch.Tensor) and self.sum.device != total.device:
                total = total.to(self.sum.device)
            self.sum = self.sum + total

    def _summary(self) -> float:
        if self.n == 0:
            return math.nan
        self._flush()
        total = self.sum.item() if isinstance(self.sum, torch.Tensor) else self.sum
        return float(total / self.n)


class AverageValueDictionaryMeter(Metric):
    """one running mean per keyword"""

    def __init__(self) -> None:
        super().__init__()
        self._meter_dicts: Dict[str, AverageValueMeter] = {}

    def _slot(self, key: str) -> AverageValueMeter:
        m = self._meter_dicts.get(key)
        if m is None:
            m = self._meter_dicts[key] = AverageValueMeter()
        return m

    def reset(self):
        for m in self._meter_dicts.values():
            m.reset()

    def _add(self, **named):
        for key, value in named.items():
            self._slot(key).add(value)

    def _summary(self):
        return {key: m.summary() for key, m in self._meter_dicts.items()}


class AverageValueListMeter(AverageValueDictionaryMeter):
    """one running mean per list position (e.g. the learning rate of every parameter group)"""

    def _add(self, list_value: Iterable[float] = None, **_):
        if not isinstance(list_value, Iterable):
            raise TypeError(f"expected an iterable of values, got {type(list_value).__name__}")
        for position, value in enumerate(list_value):
            self._slot(str(position)).add(value)
