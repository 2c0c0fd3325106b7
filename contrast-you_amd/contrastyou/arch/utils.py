"""Feature taps on named U-Net blocks (the interface the hooks use from contrastyou/arch/utils.py:
17-151): a tap is a forward hook on `model.get_module(name)` that, while switched on, remembers the
block's output tensor of every forward call -- with its autograd history -- until `clear()`.

The HIP-graph path (cyhip/graphed.py) drives the same hooks by calling them with the graph's output
tensors, so a tap never needs to know how the pass was executed.
"""
from __future__ import annotations

from contextlib import ExitStack, contextmanager
from typing import Iterator, List, Optional, Sequence, Union

import torch
from torch.nn import Module, Parameter

__all__ = ["get_requires_grad", "get_bn_track", "SingleFeatureExtractor", "FeatureExtractor"]

_MAX_PENDING_CALLS = 5  # a tap that is never cleared is a bug in the caller: fail instead of leaking


def get_requires_grad(input_: Union[Parameter, Module]) -> bool:
    """requires_grad of a parameter, or of a block (read from its first parameter: blocks are
    frozen / unfrozen as a whole by UNet.switch_grad)"""
    if isinstance(input_, Parameter):
        return input_.requires_grad
    if isinstance(input_, Module):
        return next(input_.parameters()).requires_grad
    raise AssertionError(type(input_))


def get_bn_track(input_: Module) -> bool:
    """track_running_stats of the first normalisation layer inside `input_`"""
    for sub in input_.modules():
        flag = getattr(sub, "track_running_stats", None)
        if flag is not None:
            return flag
    raise RuntimeError(f"BN module not found in {input_}")


class _FeatureCollector:
    """the hook object: call signature of torch forward hooks, storage in call order"""

    def __init__(self, max_limit: int = _MAX_PENDING_CALLS) -> None:
        self._max = max_limit
        self._on = False
        self._outputs: List[torch.Tensor] = []

    def __call__(self, _module, _inputs, output):
        if self._on:
            self._outputs.append(output)
            if len(self._outputs) >= self._max:
                raise RuntimeError(f"You may forget to call clear as this hook has registered data from "
                                   f"{len(self._outputs)} forward passes.")

    @property
    def feature(self) -> List[torch.Tensor]:
        return self._outputs

    @property
    def enable(self) -> bool:
        return self._on

    def set_enable(self, enable=True):
        self._on = bool(enable)

    def clear(self):
        self._outputs = []


class SingleFeatureExtractor:
    """tap on one block; `feature()` = all remembered outputs concatenated along the batch axis"""

    def __init__(self, model, feature_name: str) -> None:
        if feature_name not in model.arch_elements:
            raise AssertionError(feature_name)
        self._model = model
        self._feature_name = feature_name
        self._collector: Optional[_FeatureCollector] = None
        self._handle = None

    # life cycle: bind() ... remove(), or `with extractor:`
    def bind(self):
        self._collector = _FeatureCollector()
        block = self._model.get_module(self._feature_name)
        self._handle = block.register_forward_hook(self._collector)

    def remove(self):
        self._handle.remove()

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, *exc):
        self.remove()

    # recording control
    def set_enable(self, enable=True):
        self._collector.set_enable(enable)

    @contextmanager
    def enable_register(self, enable=True):
        before = self._collector.enable
        self._collector.set_enable(enable)
        try:
            yield
        finally:
            self._collector.set_enable(before)

    def clear(self):
        self._collector.clear()

    def feature(self) -> torch.Tensor:
        outs = self._collector.feature
        if not outs:
            raise RuntimeError("no feature has been recorded.")
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def tail(self, rows: int) -> torch.Tensor:
        """`feature()[-rows:]`, built from the last recordings alone when they hold exactly `rows`
        rows (the unlabeled pass of a two-stage step): earlier recordings are then neither copied nor
        tied into the autograd graph of the result"""
        outs = self._collector.feature
        if not outs:
            raise RuntimeError("no feature has been recorded.")
        have, first = 0, len(outs)
        while first > 0 and have < rows:
            first -= 1
            have += outs[first].shape[0]
        if have != rows:
            return self.feature()[-rows:]
        picked = outs[first:]
        return picked[0] if len(picked) == 1 else torch.cat(picked, dim=0)


class FeatureExtractor:
    """several taps driven together; iterating yields their features in the order of the names"""

    def __init__(self, model, feature_names: Union[str, Sequence[str]]):
        names = [feature_names] if isinstance(feature_names, str) else list(feature_names)
        self._feature_names = names
        self._taps = [SingleFeatureExtractor(model, n) for n in names]

    def _each(self, method: str, *args, **kwargs):
        for tap in self._taps:
            getattr(tap, method)(*args, **kwargs)

    def bind(self):
        self._each("bind")

    def remove(self):
        self._each("remove")

    def clear(self):
        self._each("clear")

    def set_enable(self, enable=True):
        self._each("set_enable", enable)

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, *exc):
        self.remove()

    @contextmanager
    def enable_register(self, enable=True):
        with ExitStack() as stack:
            for tap in self._taps:
                stack.enter_context(tap.enable_register(enable=enable))
            yield

    def features(self) -> Iterator[torch.Tensor]:
        return (tap.feature() for tap in self._taps)

    __iter__ = features

    def named_features(self) -> Iterator:
        return zip(self._feature_names, self.features())
