"""Output-consistency regulariser between the two unlabeled views (what semi_seg/hooks/consistency.py:
10-38 computes): the transformed prediction of the plain view is the fixed target of the prediction
on the transformed view,

    loss = weight * mean((softmax(unlabeled_logits_tf).detach() - softmax(unlabeled_tf_logits))^2)

evaluated by one fused HIP pass over both logit maps (cyhip.functions.SoftmaxMSEFn) instead of
two softmax kernels and an MSE kernel.
"""
from __future__ import annotations

from contrastyou.hooks.base import EpocherHook, TrainerHook
from contrastyou.meters import AverageValueMeter, MeterInterface
from cyhip.functions import SoftmaxMSEFn


class _ConsistencyEpocherHook(EpocherHook):
    """per-epoch worker: owns the `loss` meter of its group"""

    def __init__(self, *, name: str, weight: float) -> None:
        super().__init__(name=name)
        self._weight = float(weight)

    def configure_meters_given_epocher(self, meters: MeterInterface):
        meters = super().configure_meters_given_epocher(meters)
        meters.register_meter("loss", AverageValueMeter())
        return meters

    def _call_implementation(self, *, unlabeled_tf_logits, unlabeled_logits_tf, **_ignored):
        target = unlabeled_logits_tf.detach()  # no gradient into the plain view
        mse = SoftmaxMSEFn.apply(target, unlabeled_tf_logits)
        self.meters["loss"].add(mse.detach())
        return mse * self._weight


class ConsistencyTrainerHook(TrainerHook):
    """stateless across epochs: nothing learnable, only the weight"""

    def __init__(self, *, name: str, weight: float):
        super().__init__(hook_name=name)
        self._weight = weight

    def __call__(self) -> _ConsistencyEpocherHook:
        return _ConsistencyEpocherHook(name=self._hook_name, weight=self._weight)
