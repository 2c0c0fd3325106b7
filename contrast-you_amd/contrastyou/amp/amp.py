"""`AMPScaler` (contrastyou/amp/amp.py:13-45): loss scaling with gradient accumulation and the
autocast context of the step.  Two mixed-precision modes:

* the reference's own: fp16 autocast + `torch.amp.GradScaler` (loss scaling, inf check, skipped steps) --
  selected by handing an ENABLED GradScaler; the U-Net kernels then run their float16 instantiation
  (`mfma_f32_32x32x16_f16`, f32 accumulation, f32 BN statistics and f32 weight / BN gradients);
* the build's default: bf16 autocast with the pass-through `BF16Scaler` (bf16 has the f32 exponent
  range: nothing to scale, no host sync per step).

The autocast dtype follows the scaler unless given explicitly (cyhip.functions.compute_dtype_for reads it)."""
from __future__ import annotations

import torch


class AMPScaler:

    def __init__(self, *, scaler, accumulate_iter: int = 1, autocast_dtype: torch.dtype = None) -> None:
        assert accumulate_iter >= 1
        self.scaler = scaler
        self._accumulate_iter = accumulate_iter
        if autocast_dtype is None:
            real = isinstance(scaler, torch.amp.GradScaler) and scaler.is_enabled()
            autocast_dtype = torch.float16 if real else torch.bfloat16
        self._autocast_dtype = autocast_dtype

    def scale_loss(self, loss):
        if self._accumulate_iter != 1:  # (x / 1 is a launch forward and one backward)
            loss = loss / self._accumulate_iter
        return self.scaler.scale(loss)

    def optimizer_step(self, optimizer, *, cur_iter: int):
        """step optimizer and scaler on the last micro-batch of an accumulation window"""
        if cur_iter % self._accumulate_iter == (self._accumulate_iter - 1):
            # data-parallel + loss scaling: the gradient mean must be taken BEFORE the scaler's inf check, so
            # that every rank sees the same infs and skips (or takes) the same step (what DDP's in-backward
            # all-reduce gives the reference); FusedRAdam.step() then does not reduce a second time
            # (only an ENABLED GradScaler inspects the gradients: the disabled one of the f32 path keeps the
            # bucket / RAdam overlap of FusedRAdam.step())
            if hasattr(optimizer, "all_reduce_grads") and isinstance(self.scaler, torch.amp.GradScaler) \
                    and self.scaler.is_enabled():
                optimizer.all_reduce_grads()
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
