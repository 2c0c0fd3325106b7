"""`AMPScaler` (contrastyou/amp/amp.py:13-45): loss scaling with gradient accumulation and the
autocast context of the step.  The reference autocasts to fp16; the MI355X build computes the
U-Net in bf16 under autocast (no loss scaling needed, the scaler may be disabled) -- see
cyhip.functions.compute_dtype_for."""
from __future__ import annotations

import torch


class AMPScaler:

    def __init__(self, *, scaler, accumulate_iter: int = 1, autocast_dtype: torch.dtype = torch.bfloat16) -> None:
        assert accumulate_iter >= 1
        self.scaler = scaler
        self._accumulate_iter = accumulate_iter
        self._autocast_dtype = autocast_dtype

    def scale_loss(self, loss):
        return self.scaler.scale(loss / self._accumulate_iter)

    def optimizer_step(self, optimizer, *, cur_iter: int):
        """step optimizer and scaler on the last micro-batch of an accumulation window"""
        if cur_iter % self._accumulate_iter == (self._accumulate_iter - 1):
            self.scaler.step(optimizer)
            self.scaler.update()

    def optimizer_zero(self, optimizer, *, cur_iter: int):
        if cur_iter % self._accumulate_iter == 0:
            optimizer.zero_grad()

    @property
    def use_mixed_train(self) -> bool:
        return self.scaler._enabled  # noqa

    @property
    def autocast(self):
        return torch.autocast(device_type="cuda", dtype=self._autocast_dtype, enabled=self.use_mixed_train)


class BF16Scaler:
    """Drop-in for torch GradScaler when the step autocasts to bf16: autocast stays enabled
    (`_enabled` is what AMPScaler.autocast reads) but nothing is scaled, no inf check, no host
    sync -- bf16 has the f32 exponent range."""

    _enabled = True

    def scale(self, loss):
        return loss

    def step(self, optimizer, *args, **kwargs):
        return optimizer.step(*args, **kwargs)

    def update(self, *args, **kwargs):
        return None

    def unscale_(self, optimizer):
        return None

    def state_dict(self):
        return {}

    def load_state_dict(self, state):
        return None
