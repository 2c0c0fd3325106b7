"""The plugin API of the training loop: persistent `TrainerHook`s create one `EpocherHook` per
epoch; the epocher calls the six before/after callbacks and sums `hook(**kwargs)` into the
regularisation loss.  Interface parity with contrastyou/hooks/base.py:20-320 of the reference
(same class names, method names, kwargs contract, unique-name enforcement).
"""
from __future__ import annotations

import typing as t
import weakref
from contextlib import nullcontext

from torch import nn

from ..nn import ModuleBase
from ..utils.utils import class_name


class HookNameExistError(Exception):
    pass


class HookNotInitializedError(Exception):
    pass


class _UniqueHookName(type):
    """metaclass: two trainer hooks of one experiment may not share `hook_name`"""
    names: t.Set[str] = set()

    def __call__(cls, *args, **kwargs):
        name = kwargs.get("hook_name")
        if name is not None:
            if name in cls.names:
                raise HookNameExistError(name)
            cls.names.add(name)
        return super().__call__(*args, **kwargs)


class TrainerHook(ModuleBase, metaclass=_UniqueHookName):

    def __init__(self, *, hook_name: str):
        super().__init__()
        self._hook_name = hook_name
        self._initialized = False

    def parameters(self, recurse: bool = True):
        """only the learnable modules' parameters join the optimizer (trainer/base.py:72-73)"""
        for m in self.learnable_modules:
            yield from m.parameters(recurse=recurse)

    @property
    def learnable_modules(self) -> t.List[nn.Module]:
        return []

    def __call__(self, **kwargs) -> "EpocherHook":
        raise NotImplementedError(f"subclass {class_name(self)} must implement __call__ function.")

    def close(self):
        pass

    def after_initialize(self):
        pass

    def register_trainer(self, trainer):
        self._initialized = True
        self.register_non_trackable_buffer("trainer", trainer)
        self.register_non_trackable_buffer("_trainer", trainer)


class CombineTrainerHook(TrainerHook):

    def __init__(self, *trainer_hook: TrainerHook):
        super().__init__(hook_name="")
        self._hooks = nn.ModuleList(trainer_hook)

    def __call__(self):
        return CombineEpochHook(*[h() for h in self._hooks])

    @property
    def learnable_modules(self):
        return self._hooks

    def close(self):
        for h in self._hooks:
            h.close()

    @property
    def trainer(self):
        for h in self._hooks:
            if h._initialized:  # noqa
                return h.trainer
        raise RuntimeError(f"{class_name(self)} not initialized yet.")

    def register_trainer(self, trainer):
        for h in self._hooks:
            h.register_trainer(trainer)

    def after_initialize(self):
        for h in self._hooks:
            h.after_initialize()


class EpocherHook:

    def __init__(self, *, name: str) -> None:
        self._name = name
        self._epocher = None
        self.meters = None
        self._epocher_init = False

    @property
    def epocher(self):
        if self._epocher_init:
            return self._epocher
        raise HookNotInitializedError(f"{self._name} not initialized yet.")

    @epocher.setter
    def epocher(self, epocher):
        self._epocher = weakref.proxy(epocher)
        self.meters = weakref.proxy(epocher.meters)
        self._epocher_init = True
        with self.meters.focus_on(self.name):
            self.configure_meters_given_epocher(self.meters)

    def configure_meters_given_epocher(self, meters):
        return meters

    # ---- entry points used by the epocher (each runs focused on this hook's meter group) ----
    def _focused(self, fn, **kwargs):
        assert self._epocher_init
        with self.context:
            return fn(**kwargs)

    def call_before_batch_update(self, **kwargs):
        return self._focused(self.before_batch_update, **kwargs)

    def call_before_forward_pass(self, **kwargs):
        return self._focused(self.before_forward_pass, **kwargs)

    def call_after_forward_pass(self, **kwargs):
        return self._focused(self.after_forward_pass, **kwargs)

    def call_before_regularization(self, **kwargs):
        return self._focused(self.before_regularization, **kwargs)

    def call_after_regularization(self, **kwargs):
        return self._focused(self.after_regularization, **kwargs)

    def call_after_batch_update(self, **kwargs):
        return self._focused(self.after_batch_update, **kwargs)

    # ---- overridables ----
    def before_batch_update(self, **kwargs):
        pass

    def before_forward_pass(self, **kwargs):
        pass

    def after_forward_pass(self, **kwargs):
        pass

    def before_regularization(self, **kwargs):
        pass

    def after_regularization(self, **kwargs):
        pass

    def after_batch_update(self, **kwargs):
        pass

    def __call__(self, **kwargs):
        return self._focused(self._call_implementation, **kwargs)

    def _call_implementation(self, **kwargs):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def name(self):
        return self._name

    @property
    def context(self):
        return self.meters.focus_on(self._name) if self.meters else nullcontext()


class CombineEpochHook(EpocherHook):
    """fan-out of the callbacks; `__call__` SUMS the member losses"""

    def __init__(self, *epocher_hook: EpocherHook) -> None:  # noqa
        self._epocher_hook = epocher_hook

    def call_before_forward_pass(self, **kwargs):
        for h in self._epocher_hook:
            h.call_before_forward_pass(**kwargs)

    def call_after_forward_pass(self, **kwargs):
        for h in self._epocher_hook:
            h.call_after_forward_pass(**kwargs)

    def call_before_regularization(self, **kwargs):
        for h in self._epocher_hook:
            h.call_before_regularization(**kwargs)

    def call_after_regularization(self, **kwargs):
        for h in self._epocher_hook:
            h.call_after_regularization(**kwargs)

    def call_before_batch_update(self, **kwargs):
        for h in self._epocher_hook:
            h.call_before_batch_update(**kwargs)

    def call_after_batch_update(self, **kwargs):
        for h in self._epocher_hook:
            h.call_after_batch_update(**kwargs)

    def __call__(self, **kwargs):
        return sum(h(**kwargs) for h in self._epocher_hook)

    def _call_implementation(self, **kwargs):
        raise NotImplementedError()

    def close(self):
        for h in self._epocher_hook:
            h.close()

    @property
    def epocher(self):
        for h in self._epocher_hook:
            return h._epocher
        return None

    @epocher.setter
    def epocher(self, epocher):
        for h in self._epocher_hook:
            h.epocher = epocher
