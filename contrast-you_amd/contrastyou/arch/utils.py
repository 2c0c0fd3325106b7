"""Forward-hook feature taps (interface of contrastyou/arch/utils.py:17-151 of the reference)."""
from __future__ import annotations

from collections import OrderedDict
from contextlib import ExitStack, contextmanager
from typing import Iterator, List, Union

import torch
from torch.nn import Module, Parameter

__all__ = ["get_requires_grad", "get_bn_track", "SingleFeatureExtractor", "FeatureExtractor"]


def get_requires_grad(input_: Union[Parameter, Module]) -> bool:
    """state of the first parameter (blocks are switched as a whole)"""
    assert isinstance(input_, (Parameter, Module)), type(input_)
    if isinstance(input_, Module):
        return next(input_.parameters()).requires_grad
    return input_.requires_grad


def get_bn_track(input_: Module) -> bool:
    for m in input_.modules():
        if hasattr(m, "track_running_stats"):
            return m.track_running_stats
    raise RuntimeError(f"BN module not found in {input_}")


class _FeatureCollector:
    """forward hook that keeps the outputs of up to `max_limit - 1` calls while enabled"""

    def __init__(self, max_limit: int = 5) -> None:
        self._limit = max_limit
        self._enable = False
        self.feature: "OrderedDict[int, torch.Tensor]" = OrderedDict()

    def __call__(self, _module, _input, result):
        if not self._enable:
            return
        self.feature[len(self.feature)] = result
        if len(self.feature) >= self._limit:
            raise RuntimeError(f"You may forget to call clear as this hook has registered data from "
                               f"{len(self.feature)} forward passes.")

    def clear(self):
        self.feature = OrderedDict()

    def set_enable(self, enable=True):
        self._enable = enable

    @property
    def enable(self):
        return self._enable


class SingleFeatureExtractor:

    def __init__(self, model, feature_name: str) -> None:
        assert feature_name in model.arch_elements, feature_name
        self._model, self._feature_name = model, feature_name
        self._feature_extractor: _FeatureCollector = None  # type: ignore
        self._hook_handler = None

    def bind(self):
        self._feature_extractor = _FeatureCollector()
        self._hook_handler = self._model.get_module(self._feature_name).register_forward_hook(
            self._feature_extractor)

    def remove(self):
        self._hook_handler.remove()

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, *args, **kwargs):
        self.remove()

    def clear(self):
        self._feature_extractor.clear()

    def feature(self):
        collected = self._feature_extractor.feature
        if len(collected) == 0:
            raise RuntimeError("no feature has been recorded.")
        return torch.cat(list(collected.values()), dim=0)

    def set_enable(self, enable=True):
        self._feature_extractor.set_enable(enable=enable)

    @contextmanager
    def enable_register(self, enable=True):
        prev = self._feature_extractor.enable
        self.set_enable(enable)
        try:
            yield
        finally:
            self.set_enable(prev)


class FeatureExtractor:

    def __init__(self, model, feature_names: Union[str, List[str]]):
        self._feature_names = (feature_names,) if isinstance(feature_names, str) else feature_names
        self._extractor_list = [SingleFeatureExtractor(model, f) for f in self._feature_names]

    def bind(self):
        for e in self._extractor_list:
            e.bind()

    def remove(self):
        for e in self._extractor_list:
            e.remove()

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.remove()

    def set_enable(self, enable=True):
        for e in self._extractor_list:
            e.set_enable(enable)

    @contextmanager
    def enable_register(self, enable=True):
        with ExitStack() as stack:
            for e in self._extractor_list:
                stack.enter_context(e.enable_register(enable=enable))
            yield

    def clear(self):
        for e in self._extractor_list:
            e.clear()

    def __iter__(self):
        for e in self._extractor_list:
            yield e.feature()

    def features(self) -> Iterator:
        return iter(self)

    def named_features(self) -> Iterator:
        yield from zip(self._feature_names, self.features())
