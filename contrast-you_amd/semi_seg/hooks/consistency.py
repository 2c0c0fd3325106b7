"""Consistency regulariser (semi_seg/hooks/consistency.py:10-38):
weight * MSE(softmax(unlabeled_logits_tf).detach(), softmax(unlabeled_tf_logits)) as one fused
HIP pass (cyhip.functions.SoftmaxMSEFn)."""
from __future__ import annotations

from contrastyou.hooks.base import EpocherHook, TrainerHook
from contrastyou.meters import AverageValueMeter, MeterInterface
from cyhip.functions import SoftmaxMSEFn


class ConsistencyTrainerHook(TrainerHook):

    def __init__(self, *, name: str, weight: float):
        super().__init__(hook_name=name)
        self._weight = weight

    def __call__(self):
        return _ConsistencyEpocherHook(name=self._hook_name, weight=self._weight)


class _ConsistencyEpocherHook(EpocherHook):
    def __init__(self, *, name: str, weight: float) -> None:
        super().__init__(name=name)
        self._weight = weight

    def configure_meters_given_epocher(self, meters: MeterInterface):
        meters = super().configure_meters_given_epocher(meters)
        meters.register_meter("loss", AverageValueMeter())
        return meters

    def _call_implementation(self, *, unlabeled_tf_logits, unlabeled_logits_tf, seed, affine_transformer, **kwargs):
        loss = SoftmaxMSEFn.apply(unlabeled_logits_tf.detach(), unlabeled_tf_logits)
        self.meters["loss"].add(loss.detach())
        return self._weight * loss
