"""The plugin API of the training loop (what contrastyou/hooks/base.py:20-320 defines in the
reference; same class names, method names and kwargs contract, so hooks written against the
reference run here):

* a `TrainerHook` lives as long as the trainer, is an `nn.Module` (its `learnable_modules` join the
  optimizer and the checkpoint) and is called once per epoch to hand out an `EpocherHook`;
* an `EpocherHook` is driven by the epocher through six stage callbacks around the batch update, the
  forward pass and the regularisation, and through `hook(**kwargs)`, whose results the epocher sums
  into the regularisation loss.  Every call runs with the epocher's meters focused on the hook's
  own group, so `self.meters["loss"]` inside a hook is that hook's meter;
* `Combine*` wrappers fan a call out to several hooks.

The stage plumbing is table-driven here: `_STAGES` names the six callbacks once, the `call_<stage>`
entry points are generated from it.
"""
from __future__ import annotations

import weakref
from contextlib import nullcontext
from typing import Iterator, List, Set

from torch import nn

from ..nn import ModuleBase
from ..utils.utils import class_name

_STAGES = ("before_batch_update", "before_forward_pass", "after_forward_pass",
           "before_regularization", "after_regularization", "after_batch_update")


class HookNameExistError(Exception):
    """two trainer hooks were given the same `hook_name`"""


class HookNotInitializedError(Exception):
    """an epocher hook was used before an epocher adopted it"""


class _UniqueHookName(type(ModuleBase)):
    """metaclass of TrainerHook: remembers every `hook_name=` ever used and refuses duplicates
    (meter groups and checkpoint entries are keyed by it)"""
    names: Set[str] = set()

    def __call__(cls, *args, **kwargs):
        wanted = kwargs.get("hook_name")
        if wanted is not None:
            if wanted in _UniqueHookName.names:
                raise HookNameExistError(wanted)
            _UniqueHookName.names.add(wanted)
        return super().__call__(*args, **kwargs)


class TrainerHook(ModuleBase, metaclass=_UniqueHookName):

    def __init__(self, *, hook_name: str):
        super().__init__()
        self._hook_name = hook_name
        self._initialized = False  # becomes True when a trainer registers the hook

    @property
    def learnable_modules(self) -> List[nn.Module]:
        """modules whose parameters the trainer hands to the optimizer (second parameter group)"""
        return []

    def parameters(self, recurse: bool = True) -> Iterator[nn.Parameter]:
        for module in self.learnable_modules:
            yield from module.parameters(recurse=recurse)

    def __call__(self, **kwargs) -> "EpocherHook":
        raise NotImplementedError(f"subclass {class_name(self)} must implement __call__ function.")

    def register_trainer(self, trainer):
        for attr in ("trainer", "_trainer"):  # plain references: not saved, not moved with .to()
            self.register_non_trackable_buffer(attr, trainer)
        self._initialized = True

    def after_initialize(self):
        """called once all hooks of a trainer are registered"""

    def close(self):
        """called when the trainer leaves its `register_hook` block"""


class CombineTrainerHook(TrainerHook):
    """several trainer hooks behind one: parameters, checkpoints and life-cycle calls fan out"""

    def __init__(self, *trainer_hook: TrainerHook):
        super().__init__(hook_name="")
        self._hooks = nn.ModuleList(trainer_hook)

    @property
    def learnable_modules(self):
        return self._hooks

    def __call__(self):
        return CombineEpochHook(*(member() for member in self._hooks))

    @property
    def trainer(self):
        for member in self._hooks:
            if member._initialized:  # noqa
                return member.trainer
        raise RuntimeError(f"{class_name(self)} not initialized yet.")

    def register_trainer(self, trainer):
        for member in self._hooks:
            member.register_trainer(trainer)

    def after_initialize(self):
        for member in self._hooks:
            member.after_initialize()

    def close(self):
        for member in self._hooks:
            member.close()


class EpocherHook:

    def __init__(self, *, name: str) -> None:
        self._name = name
        self._epocher = None
        self._epocher_init = False
        self.meters = None

    # ---- adoption by an epocher ---------------------------------------------------------------
    @property
    def epocher(self):
        if not self._epocher_init:
            raise HookNotInitializedError(f"{self._name} not initialized yet.")
        return self._epocher

    @epocher.setter
    def epocher(self, epocher):
        self._epocher = weakref.proxy(epocher)  # the epocher owns the hook, not the other way round
        self.meters = weakref.proxy(epocher.meters)
        self._epocher_init = True
        with self.context:
            self.configure_meters_given_epocher(self.meters)

    def configure_meters_given_epocher(self, meters):
        """register this hook's meters (runs focused on the hook's meter group)"""
        return meters

    @property
    def name(self):
        return self._name

    @property
    def context(self):
        return self.meters.focus_on(self._name) if self.meters else nullcontext()

    def _focused(self, fn, **kwargs):
        assert self._epocher_init
        with self.context:
            return fn(**kwargs)

    # ---- what subclasses fill in -----------------------------------------------------------------
    def _call_implementation(self, **kwargs):
        raise NotImplementedError

    def __call__(self, **kwargs):
        return self._focused(self._call_implementation, **kwargs)

    def close(self):
        """end of the epoch"""


def _noop_stage(self, **kwargs):
    return None


def _make_entry_point(stage: str):
    def entry(self, **kwargs):
        return self._focused(getattr(self, stage), **kwargs)

    entry.__name__ = "call_" + stage
    entry.__doc__ = f"epocher-side entry: run `{stage}` focused on this hook's meter group"
    return entry


for _stage in _STAGES:  # before_*/after_* default to no-ops, call_<stage> wraps them
    setattr(EpocherHook, _stage, _noop_stage)
    setattr(EpocherHook, "call_" + _stage, _make_entry_point(_stage))


class CombineEpochHook(EpocherHook):
    """fan-out over member hooks; `__call__` SUMS the member losses"""

    def __init__(self, *epocher_hook: EpocherHook) -> None:  # noqa: no own name / meters
        self._epocher_hook = epocher_hook

    @property
    def epocher(self):
        return self._epocher_hook[0]._epocher if self._epocher_hook else None

    @epocher.setter
    def epocher(self, epocher):
        for member in self._epocher_hook:
            member.epocher = epocher

    def __call__(self, **kwargs):
        return sum(member(**kwargs) for member in self._epocher_hook)

    def _call_implementation(self, **kwargs):
        raise NotImplementedError()

    def close(self):
        for member in self._epocher_hook:
            member.close()


def _make_fan_out(stage: str):
    def fan_out(self, **kwargs):
        for member in self._epocher_hook:
            getattr(member, "call_" + stage)(**kwargs)

    fan_out.__name__ = "call_" + stage
    return fan_out


for _stage in _STAGES:
    setattr(CombineEpochHook, "call_" + _stage, _make_fan_out(_stage))
del _stage
