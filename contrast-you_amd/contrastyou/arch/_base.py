"""Block-name arithmetic shared by hooks and trainers, and the protocol a segmentation network
offers to them (the `_Network` protocol, `sort_arch`, `_check_params`, `_complete_arch_start2end`
of contrastyou/arch/_base.py:8-84).  Everything reduces to the position of a name in
`model.arch_elements`.
"""
from __future__ import annotations

from typing import ContextManager, Dict, List, Optional, Protocol, Sequence

from torch import nn


def arch_order(name: str, *, model) -> int:
    """0-based position of a block in the forward order of `model` (KeyError for unknown names)"""
    order = list(model.arch_elements)
    if name not in order:
        raise KeyError(name)
    return order.index(name)


def sort_arch(name_list: List[str], reverse: bool = False, *, model) -> List[str]:
    """`name_list` in forward order (or backward with reverse=True)"""
    return sorted(name_list, key=lambda block: arch_order(block, model=model), reverse=reverse)


def _complete_arch_start2end(start: str, end: str, include_start=True, include_end=True, *, model) -> List[str]:
    """the blocks from `start` to `end` in forward order, ends included or not"""
    first, last = arch_order(start, model=model), arch_order(end, model=model)
    assert first <= last, (start, end)
    first += 0 if include_start else 1
    last -= 0 if include_end else 1
    return list(model.arch_elements)[first: last + 1]


def _check_params(start, end, include_start, include_end, *, model) -> None:
    """argument validation of UNet.switch_grad / switch_bn_track: an open end must be inclusive,
    names must exist, and `start` may not lie behind `end`"""
    if start is None and not include_start:
        raise ValueError("include_start should be True given start=None")
    if end is None and not include_end:
        raise ValueError("include_end should be True given end=None")
    named = [v for v in (start, end) if isinstance(v, str)]
    for block in named:
        if block not in model.arch_elements:
            raise ValueError(block)
    if len(named) == 2 and arch_order(start, model=model) > arch_order(end, model=model):
        raise ValueError((start, end))


class _Network(Protocol):
    """what hooks and trainers rely on (see contrastyou.arch.unet.UNet)"""
    arch_elements: Sequence[str]                  # all block names in forward order
    encoder_names: Sequence[str]
    decoder_names: Sequence[str]
    layer_dimension: Dict[str, Optional[int]]     # channel multiplier per block

    @property
    def num_classes(self) -> int: ...

    def get_channel_dim(self, name: str) -> int: ...

    def get_module(self, name: str) -> nn.Module: ...

    def switch_grad(self, **kwargs) -> ContextManager: ...

    def switch_bn_track(self, **kwargs) -> ContextManager: ...
