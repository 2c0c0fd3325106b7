"""`EpocherBase`: one epoch of work -- a model, a meter registry, a progress indicator and the
per-epoch hooks (the role of contrastyou/epochers/base.py:43-168).  Life cycle, as in the reference:

    epocher = SomeEpocher(model=..., num_batches=..., scaler=..., accumulate_iter=...)
    epocher.init(trainer)                      # meters + indicator exist from here on
    with epocher.register_hook(*epoch_hooks):  # optional
        epocher.run()
    epocher.get_metric(), epocher.get_score()
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
    """what the epochers need from the reference's customised tqdm: iteration over the batch
    indices, a description, the latest statistics; silent unless asked to be verbose"""

    def __init__(self, n: int, disable: bool = True):
        self._n, self._disable = n, disable
        self._desc, self._last = "", None

    def __iter__(self):
        return iter(range(self._n))

    def set_desc_from_epocher(self, epocher):
        self._desc = f"{class_name(epocher)} {epocher.cur_epoch}"

    def set_postfix_statics2(self, report, force_update=False):
        self._last = report

    def log_result(self):
        if self._last is not None and not self._disable:
            print(self._desc, dict(self._last))

    def close(self):
        pass


class EpocherBase(AMPScaler, DDPMixin, metaclass=ABCMeta):
    meter_focus = "tra"  # name of the epocher's own meter group

    def __init__(self, *, model: nn.Module, num_batches: int, cur_epoch=0, device="cpu", scaler,
                 accumulate_iter: int, **kwargs) -> None:
        super().__init__(scaler=scaler, accumulate_iter=accumulate_iter)
        self._model = model
        self._device = torch.device(device)
        self._num_batches, self._cur_epoch = num_batches, cur_epoch
        self._trainer = None
        self._hooks: List = []
        self._initialized = False
        self.meters: MeterInterface = None  # type: ignore[assignment]
        self.indicator: _Indicator = None   # type: ignore[assignment]
        self.verbose = False

    # ---- read-only facts ------------------------------------------------------------------------
    device = property(lambda self: self._device)
    cur_epoch = property(lambda self: self._cur_epoch)
    num_batches = property(lambda self: self._num_batches)

    # ---- the owning trainer (held weakly) ---------------------------------------------------------
    @property
    def trainer(self):
        if self._trainer is None:
            raise TrainerNotSetError(f"{self.__class__.__name__} should call `set_trainer` first")
        return self._trainer

    @trainer.setter
    def trainer(self, trainer):
        self._trainer = weakref.proxy(trainer)

    set_trainer = trainer.fset

    # ---- life cycle -------------------------------------------------------------------------------
    def init(self, trainer=None):
        self.meters = MeterInterface(default_focus=self.meter_focus)
        self.configure_meters(self.meters)
        self.indicator = _Indicator(self._num_batches, disable=not (self.verbose and self.on_master))
        if trainer is not None:
            self.trainer = trainer
        self._initialized = True

    @abstractmethod
    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters.register_meter("lr", AverageValueListMeter())
        return meters

    @contextmanager
    def register_hook(self, *hook):
        assert self._initialized, f"{class_name(self)} must be initialized by calling {class_name(self)}.init()."
        for h in hook:
            h.epocher = self  # wires meters and lets the hook register its own
            self._hooks.append(h)
        yield
        self.close_hook()

    def close_hook(self):
        for h in self._hooks:
            h.close()

    @abstractmethod
    def _run(self, **kwargs):
        raise NotImplementedError()

    def run(self, **kwargs):
        self.to(self.device)
        self.indicator.set_desc_from_epocher(self)
        with self.meters:  # reset on entry, join on exit
            result = self._run(**kwargs)
        self.indicator.close()
        self.indicator.log_result()
        return result

    def to(self, device: Union[torch.device, str] = torch.device("cpu")):
        device = torch.device(device)
        for member in vars(self).values():
            if isinstance(member, nn.Module):
                member.to(device)
        self._device = device

    # ---- results ----------------------------------------------------------------------------------
    def get_metric(self) -> Dict[str, Dict[str, float]]:
        return dict(self.meters.statistics())

    def get_score(self) -> float:
        raise NotImplementedError()
