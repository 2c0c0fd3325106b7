"""`EpocherBase` (contrastyou/epochers/base.py:43-168): one epoch of work with meters, a progress
indicator and per-epoch hooks.  Same constructor kwargs and lifecycle
(`init()` -> `with register_hook(*hooks): run()` -> `get_metric()/get_score()`).
"""
from __future__ import annotations

import weakref
from abc import ABCMeta, abstractmethod
from contextlib import contextmanager
from typing import Dict, List, Union

import torch
from torch import nn

from ..amp import AMPScaler, DDPMixin
from ..meters import AverageValueListMeter, MeterInterface
from ..utils.utils import class_name


class TrainerNotSetError(Exception):
    pass


class _Indicator:
    """minimal stand-in for the reference's customised tqdm: iterates range(n) and remembers the
    last statistics; prints nothing unless `verbose`."""

    def __init__(self, n: int, disable: bool = True):
        self._n, self._disable, self._desc, self._last = n, disable, "", None

    def __iter__(self):
        return iter(range(self._n))

    def set_desc_from_epocher(self, epocher):
        self._desc = f"{class_name(epocher)} {epocher.cur_epoch}"

    def set_postfix_statics2(self, report, force_update=False):
        self._last = report

    def close(self):
        pass

    def log_result(self):
        if not self._disable and self._last is not None:
            print(self._desc, dict(self._last))


class EpocherBase(AMPScaler, DDPMixin, metaclass=ABCMeta):
    meter_focus = "tra"

    def __init__(self, *, model: nn.Module, num_batches: int, cur_epoch=0, device="cpu", scaler,
                 accumulate_iter: int, **kwargs) -> None:
        super().__init__(scaler=scaler, accumulate_iter=accumulate_iter)
        self._initialized = False
        self._model = model
        self._device = device if isinstance(device, torch.device) else torch.device(device)
        self._num_batches = num_batches
        self._cur_epoch = cur_epoch
        self._trainer = None
        self._hooks: List = []
        self.meters = None
        self.indicator = None
        self.verbose = False

    @property
    def trainer(self):
        if self._trainer is not None:
            return self._trainer
        raise TrainerNotSetError(f"{self.__class__.__name__} should call `set_trainer` first")

    @trainer.setter
    def trainer(self, trainer):
        self._trainer = weakref.proxy(trainer)

    def set_trainer(self, trainer):
        self._trainer = weakref.proxy(trainer)

    def init(self, trainer=None):
        self.meters = MeterInterface(default_focus=self.meter_focus)
        self.configure_meters(self.meters)
        self.indicator = _Indicator(self._num_batches, disable=not (self.on_master and self.verbose))
        self._initialized = True
        if trainer is not None:
            self.trainer = trainer

    @contextmanager
    def register_hook(self, *hook):
        assert self._initialized, f"{class_name(self)} must be initialized by calling {class_name(self)}.init()."
        for h in hook:
            self._hooks.append(h)
            h.epocher = self
        yield
        self.close_hook()

    def close_hook(self):
        for h in self._hooks:
            h.close()

    @abstractmethod
    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters.register_meter("lr", AverageValueListMeter())
        return meters

    @abstractmethod
    def _run(self, **kwargs):
        raise NotImplementedError()

    def run(self, **kwargs):
        self.to(self.device)
        self.indicator.set_desc_from_epocher(self)
        with self.meters:
            result = self._run(**kwargs)
        self.indicator.close()
        self.indicator.log_result()
        return result

    def get_metric(self) -> Dict[str, Dict[str, float]]:
        return dict(self.meters.statistics())

    def get_score(self) -> float:
        raise NotImplementedError()

    def to(self, device: Union[torch.device, str] = torch.device("cpu")):
        device = torch.device(device) if isinstance(device, str) else device
        for m in self.__dict__.values():
            if isinstance(m, nn.Module):
                m.to(device)
        self._device = device

    @property
    def device(self):
        return self._device

    @property
    def cur_epoch(self):
        return self._cur_epoch

    @property
    def num_batches(self):
        return self._num_batches
