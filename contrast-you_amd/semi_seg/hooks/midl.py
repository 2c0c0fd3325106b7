"""IIC on the segmentation outputs themselves (semi_seg/hooks/midl.py:18-54): mutual information
between softmax(logits of the transformed view) and the transformed softmax of the original view,
`IIDSegmentationLoss(padding=0, lamda=mi_lambda)`.
"""
from __future__ import annotations

from contrastyou.hooks.base import EpocherHook, TrainerHook
from contrastyou.losses.discreteMI import IIDSegmentationLoss
from contrastyou.losses.kl import Entropy
from contrastyou.meters import AverageValueMeter
from cyhip.functions import GroupSoftmaxFn

entropy_criterion = Entropy(reduction="none", eps=1e-8)


class IIDSegmentationTrainerHook(TrainerHook):

    def __init__(self, *, hook_name: str = "midl_hook", weight: float = 1.0, mi_lambda=1.0) -> None:
        super().__init__(hook_name=hook_name)
        self._weight = weight
        self._mi_lambda = mi_lambda

    def __call__(self):
        return _IIDSegmentationEpochHook(name=self._hook_name, weight=self._weight, mi_lambda=self._mi_lambda)


def _softmax_map(logits):
    """softmax over channels of an [n,K,H,W] logit map on the HIP row-softmax kernel"""
    n, k, h, w = logits.shape
    flat = logits.float().permute(0, 2, 3, 1).reshape(n * h * w, k)
    return GroupSoftmaxFn.apply(flat, 1, k, 1.0)[0].view(n, h, w, k).permute(0, 3, 1, 2)


class _IIDSegmentationEpochHook(EpocherHook):

    def __init__(self, *, name: str, weight: float, mi_lambda=1.0) -> None:
        super().__init__(name=name)
        self._weight = weight
        self._criterion = IIDSegmentationLoss(padding=0, lamda=mi_lambda)

    def configure_meters_given_epocher(self, meters):
        meters.register_meter("mi", AverageValueMeter())

    def _call_implementation(self, *, unlabeled_tf_logits, unlabeled_logits_tf, **kwargs):
        loss = self._criterion(_softmax_map(unlabeled_tf_logits), _softmax_map(unlabeled_logits_tf))
        self.meters["mi"].add(loss.detach())
        return loss * self._weight
