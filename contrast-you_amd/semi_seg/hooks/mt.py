"""Mean-teacher regulariser (semi_seg/hooks/mt.py:49-207): an EMA teacher copy of the model
predicts the unlabeled batch (no grad; train-mode BN, or -- with update_bn, where the EMA also tracks
the running statistics -- every teacher BatchNorm in eval mode, mt.py:162-166), its logits go through
the same affine map, and weight * MSE(softmax(teacher_tf) [hard_clip: its arg-max one-hot, mt.py:190-192],
softmax(student_tf)) is added; after the step the teacher is updated by `EMAUpdater` = cy_ema_update
over every parameter."""
from __future__ import annotations

from copy import deepcopy

import torch
from torch import nn

from contrastyou.hooks.base import EpocherHook, TrainerHook
from contrastyou.meters import AverageValueMeter, MeterInterface
from cyhip import ops
from cyhip.functions import SoftmaxMSEFn, bump_weights_epoch


class EMAUpdater:
    def __init__(self, alpha=0.999, justify_alpha=True, weight_decay=1e-5, update_bn=False) -> None:
        self._alpha, self._weight_decay = alpha, weight_decay
        self._update_bn, self._justify_alpha = update_bn, justify_alpha
        self._global_step = 0

    @torch.no_grad()
    def __call__(self, ema_model: nn.Module, student_model: nn.Module):
        alpha = min(1 - 1 / (self._global_step + 1), self._alpha) if self._justify_alpha else self._alpha
        for e, s in zip(ema_model.parameters(), student_model.parameters()):
            ops.ema_update(e.data, s.data, alpha, self._weight_decay)
        if self._update_bn:
            for (name, eb), (_, sb) in zip(ema_model.named_buffers(), student_model.named_buffers()):
                if "running_mean" in name or "running_var" in name:
                    ops.ema_update(eb.data, sb.data, alpha, self._weight_decay)
        bump_weights_epoch()  # teacher weights changed outside torch's version counter
        self._global_step += 1


def detach_model(model: nn.Module):
    for p in model.parameters():
        p.detach_()
        p.requires_grad_(False)


class MeanTeacherTrainerHook(TrainerHook):

    def __init__(self, *, name: str, model: nn.Module, weight: float, alpha: float = 0.999,
                 weight_decay: float = 1e-5, update_bn=False, num_teachers: int = 1, hard_clip=False):
        super().__init__(hook_name=name)
        if num_teachers > 1:
            raise RuntimeError(f"Current version only support one Teacher, given {num_teachers} Teachers.")
        self._weight = weight
        self._updater = EMAUpdater(alpha=alpha, weight_decay=weight_decay, update_bn=update_bn)
        self._teacher_model = deepcopy(model)
        self._hard_clip = hard_clip
        detach_model(self._teacher_model)

    def __call__(self):
        return _MeanTeacherEpocherHook(name=self._hook_name, weight=self._weight, model=self.trainer._model,
                                       teacher_model=self._teacher_model, updater=self._updater,
                                       hard_clip=self._hard_clip)

    @property
    def teacher_model(self):
        return self._teacher_model

    @property
    def learnable_modules(self):
        return []


class _MeanTeacherEpocherHook(EpocherHook):
    def __init__(self, *, name: str, weight: float, model, teacher_model, updater: EMAUpdater,
                 hard_clip: bool = False) -> None:
        super().__init__(name=name)
        self._weight, self._model, self._teacher_model, self._updater = weight, model, teacher_model, updater
        self._hard_clip = hard_clip
        self._teacher_model.train()
        if updater._update_bn:  # the EMA'd running statistics are the ones to use: freeze every BN to eval()
            for m in self._teacher_model.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()

    def configure_meters_given_epocher(self, meters: MeterInterface):
        meters = super().configure_meters_given_epocher(meters)
        meters.register_meter("loss", AverageValueMeter())
        return meters

    def _call_implementation(self, *, unlabeled_image, unlabeled_tf_logits, seed, affine_transformer, **kwargs):
        with torch.no_grad():
            teacher_logits = self._teacher_model(unlabeled_image)
            teacher_logits_tf = affine_transformer(teacher_logits)
            if self._hard_clip:
                # one-hot of the teacher's arg-max class, expressed as logits whose softmax IS that one-hot
                # (exp(-1e4) == 0 in f32), so that the fused softmax-pair MSE kernel applies unchanged
                C = teacher_logits_tf.shape[1]
                hard = torch.nn.functional.one_hot(teacher_logits_tf.argmax(1), C).movedim(-1, 1)
                teacher_logits_tf = hard.to(torch.float32) * 1e4
        loss = SoftmaxMSEFn.apply(teacher_logits_tf.detach(), unlabeled_tf_logits)
        self.meters["loss"].add(loss.detach())
        return self._weight * loss

    def after_batch_update(self, **kwargs):
        self._updater(ema_model=self._teacher_model, student_model=self._model)
