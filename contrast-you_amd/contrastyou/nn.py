"""`ModuleBase`: nn.Module plus (a) `Buffer`-wrapped plain-python state that is saved with the
checkpoint and (b) non-trackable references that are NOT saved (e.g. a hook pointing at the
model it taps).  Same public API and checkpoint schema as contrastyou/nn.py:13-168 of the
reference -- `{"module_state", "buffer_state", "other_state"}` -- without its dependency on the
removed `torch._six`.
"""
from __future__ import annotations

from collections import OrderedDict, namedtuple
from typing import Any, Dict, List, Set, Union

import torch
from torch import nn
from torch.optim import Optimizer

__all__ = ["ModuleBase", "Buffer", "NoTrackable"]


class Buffer:
    """marks a value as persistent (checkpointed) python state of a ModuleBase"""

    def __init__(self, data=None):
        if isinstance(data, (nn.Module, Optimizer, torch.optim.lr_scheduler.LRScheduler)):
            raise ValueError(f"cannot wrap a {data.__class__.__name__} in a Buffer")
        self.data = data

    def __repr__(self):
        return f"{self.__class__.__name__}({self.data})"


class NoTrackable:
    """marks a value as a plain reference that must not be saved or moved"""

    def __init__(self, data) -> None:
        self.data = data

    def __repr__(self):
        return f"{self.__class__.__name__}({self.data})"


def _check_name(name: str):
    if not isinstance(name, str):
        raise TypeError(f"buffer name should be a string. Got {type(name)}")
    if "." in name:
        raise KeyError("buffer name can't contain \".\"")
    if name == "":
        raise KeyError("buffer name can't be empty string \"\"")


class ModuleBase(nn.Module):

    def __init__(self) -> None:
        super().__init__()
        self._persist_buffer: "OrderedDict[str, Any]" = OrderedDict()
        self._non_trackable_buffer: Set[str] = set()

    # -- attribute routing -------------------------------------------------
    def _forget(self, name: str):
        for d in (self.__dict__, self._buffers, self._modules, self._persist_buffer):
            d.pop(name, None)
        self._non_persistent_buffers_set.discard(name)
        self._non_trackable_buffer.discard(name)

    def __setattr__(self, name, value):
        if isinstance(value, Buffer):
            self._forget(name)
            self.register_persist_buffer(name, value.data)
        elif "_persist_buffer" in self.__dict__ and name in self._persist_buffer:
            self._persist_buffer[name] = value
        elif isinstance(value, NoTrackable):
            self._forget(name)
            self.register_non_trackable_buffer(name, value.data)
        elif "_non_trackable_buffer" in self.__dict__ and name in self._non_trackable_buffer:
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def __getattr__(self, item):
        pb = self.__dict__.get("_persist_buffer")
        if pb is not None and item in pb:
            return pb[item]
        return super().__getattr__(item)

    def __delattr__(self, item):
        if item in self._non_trackable_buffer:
            self._non_trackable_buffer.remove(item)
            object.__delattr__(self, item)
        elif item in self._persist_buffer:
            del self._persist_buffer[item]
        else:
            super().__delattr__(item)

    def register_persist_buffer(self, name: str, data: Any):
        if "_persist_buffer" not in self.__dict__:
            raise AttributeError("cannot assign buffer before ModuleBase.__init__() call")
        _check_name(name)
        if hasattr(self, name) and name not in self._persist_buffer:
            raise KeyError(f"attribute '{name}' already exists")
        self._persist_buffer[name] = data

    def register_non_trackable_buffer(self, name: str, module: Any):
        if "_non_trackable_buffer" not in self.__dict__:
            raise AttributeError("cannot assign buffer before ModuleBase.__init__() call")
        _check_name(name)
        if hasattr(self, name) and name not in self._non_trackable_buffer:
            raise KeyError(f"attribute '{name}' already exists")
        self._non_trackable_buffer.add(name)
        object.__setattr__(self, name, module)

    # -- checkpoint schema -------------------------------------------------
    def _other_state_dict(self):
        return {k: v.state_dict() for k, v in self.__dict__.items()
                if k not in self._non_trackable_buffer and hasattr(v, "state_dict") and callable(v.state_dict)}

    def state_dict(self, *args, **kwargs):
        return OrderedDict(module_state=super().state_dict(*args, **kwargs),
                           buffer_state=self._persist_buffer.copy(),
                           other_state=self._other_state_dict())

    def load_state_dict(self, state_dict: Dict[str, Any], strict=True):
        if "module_state" not in state_dict:
            raise ValueError("Missing module_state in state_dict")
        inc = super().load_state_dict(state_dict["module_state"], strict)
        buf = state_dict.get("buffer_state", {})
        missing = [k for k in self._persist_buffer if k not in buf]
        unexpected = [k for k in buf if k not in self._persist_buffer]
        for k in self._persist_buffer:
            if k in buf:
                self._persist_buffer[k] = buf[k]
        other = state_dict.get("other_state", {})
        mine = self._other_state_dict()
        missing += [k for k in mine if k not in other]
        unexpected += [k for k in other if k not in mine]
        for k in mine:
            if k in other:
                getattr(self, k).load_state_dict(other[k])
        if strict and (missing or unexpected):
            msgs: List[str] = []
            if missing:
                msgs.append("Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing))
            if unexpected:
                msgs.append("Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected))
            raise RuntimeError(f"Error(s) in loading state_dict for {self.__class__.__name__}:\n\t" + "\n\t".join(msgs))
        return _IncompatibleKeys(list(inc.missing_keys) + missing, list(inc.unexpected_keys) + unexpected)

    def to(self, device: Union[str, torch.device], **kwargs):
        for k, v in self.__dict__.items():
            if k in self._non_trackable_buffer:
                continue
            if isinstance(v, Optimizer):
                for st in v.state.values():
                    for kk, t in list(st.items()):
                        if isinstance(t, torch.Tensor):
                            st[kk] = t.to(device)
        return super().to(device=device, **kwargs)


class _IncompatibleKeys(namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])):
    def __repr__(self):
        if not self.missing_keys and not self.unexpected_keys:
            return "<All keys matched successfully>"
        return super().__repr__()

    __str__ = __repr__
