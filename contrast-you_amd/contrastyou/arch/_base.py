"""Architecture protocol + ordering helpers shared by hooks and trainers.

Interface parity with contrastyou/arch/_base.py:8-84 of the reference (`_Network` protocol,
`sort_arch`, `_check_params`, `_complete_arch_start2end`); the implementation here is a plain
index lookup over `model.arch_elements`.
"""
from __future__ import annotations

from typing import ContextManager, Dict, List, Optional, Protocol, Sequence

from torch import nn


class _Network(Protocol):
    encoder_names: Sequence[str]
    decoder_names: Sequence[str]
    arch_elements: Sequence[str]
    layer_dimension: Dict[str, Optional[int]]

    def switch_grad(self, **kwargs) -> ContextManager: ...

    def switch_bn_track(self, **kwargs) -> ContextManager: ...

    @property
    def num_classes(self) -> int: ...

    def get_channel_dim(self, name: str) -> int: ...

    def get_module(self, name: str) -> nn.Module: ...


def arch_order(name: str, *, model) -> int:
    """position of a block name in model.arch_elements (KeyError if unknown)"""
    try:
        return list(model.arch_elements).index(name)
    except ValueError:
        raise KeyError(name) from None


def sort_arch(name_list: List[str], reverse: bool = False, *, model) -> List[str]:
    return sorted(name_list, key=lambda n: arch_order(n, model=model), reverse=reverse)


def _check_params(start, end, include_start, include_end, *, model) -> None:
    if start is None and include_start is False:
        raise ValueError("include_start should be True given start=None")
    if end is None and include_end is False:
        raise ValueError("include_end should be True given end=None")
    for v in (start, end):
        if isinstance(v, str) and v not in model.arch_elements:
            raise ValueError(v)
    if isinstance(start, str) and isinstance(end, str):
        if arch_order(start, model=model) > arch_order(end, model=model):
            raise ValueError((start, end))


def _complete_arch_start2end(start: str, end: str, include_start=True, include_end=True, *, model) -> List[str]:
    """all block names between start and end in architecture order"""
    lo, hi = arch_order(start, model=model), arch_order(end, model=model)
    assert lo <= hi, (start, end)
    if not include_start:
        lo += 1
    if not include_end:
        hi -= 1
    return list(model.arch_elements)[lo: hi + 1]
