"""`GradualWarmupScheduler`: linear warm-up of every group's lr from `base_lr` to
`base_lr * multiplier` over `total_epoch` epochs, then hand-over to `after_scheduler` whose
base lrs are rescaled by `multiplier` at the hand-over.

Same constructor, `get_lr()` law, hand-over rule and `state_dict` nesting as
contrastyou/optim/scheduler.py:19-104 (config/base.yaml:14-17: multiplier 300, warmup_max 10,
followed by CosineAnnealingLR(T_max=max_epoch-warmup_max, eta_min=1e-7), trainer/base.py:77-89).
The scheduler only writes `param_group["lr"]`; FusedRAdam reads it on every step.
"""
from __future__ import annotations

from torch.optim.lr_scheduler import LRScheduler, ReduceLROnPlateau

__all__ = ["GradualWarmupScheduler"]


class GradualWarmupScheduler(LRScheduler):

    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler: LRScheduler = None):
        self.multiplier = multiplier
        if self.multiplier <= 1.0:
            raise ValueError("multiplier should be greater than 1.")
        self.total_epoch = total_epoch
        self.after_scheduler = after_scheduler
        self.finished = False
        super().__init__(optimizer)

    def state_dict(self):
        state = {k: v for k, v in self.__dict__.items() if k not in ("optimizer", "after_scheduler")}
        if self.after_scheduler:
            state["after_scheduler"] = self.after_scheduler.state_dict()
        return state

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        after = state_dict.pop("after_scheduler", None)
        state_dict.pop("optimizer", None)
        self.__dict__.update(state_dict)
        if after and self.after_scheduler:
            self.after_scheduler.load_state_dict(after)

    def _warm(self):
        f = (self.multiplier - 1.0) * self.last_epoch / self.total_epoch + 1.0
        return [b * f for b in self.base_lrs]

    def get_lr(self):
        if self.last_epoch > self.total_epoch:
            if self.after_scheduler:
                if not self.finished:
                    self.after_scheduler.base_lrs = [b * self.multiplier for b in self.base_lrs]
                    self.finished = True
                prev = getattr(self.after_scheduler, "_get_lr_called_within_step", False)
                self.after_scheduler._get_lr_called_within_step = True
                try:
                    return self.after_scheduler.get_lr()
                finally:
                    self.after_scheduler._get_lr_called_within_step = prev
            return [b * self.multiplier for b in self.base_lrs]
        return self._warm()

    def step_ReduceLROnPlateau(self, metrics, epoch=None):  # noqa: N802 (reference name)
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch if epoch != 0 else 1
        if self.last_epoch <= self.total_epoch:
            for group, lr in zip(self.optimizer.param_groups, self._warm()):
                group["lr"] = lr
        else:
            self.after_scheduler.step(metrics)

    def step(self, epoch=None, metrics=None):
        if isinstance(self.after_scheduler, ReduceLROnPlateau):
            return self.step_ReduceLROnPlateau(metrics, epoch)
        if self.finished and self.after_scheduler:
            if epoch is None:
                self.after_scheduler.step()
            else:
                self.after_scheduler.step(epoch - self.total_epoch)
            self._last_lr = [g["lr"] for g in self.optimizer.param_groups]
            return None
        if epoch is None:
            return super().step()
        return super().step(epoch)
