"""Discrete mutual-information (IIC) regulariser on a feature map
(semi_seg/hooks/discretemi.py:16-116): cluster head with several sub-heads on the features of the
two unlabeled views, IIDLoss (encoder features, [n,k]) or IIDSegmentationLoss (decoder features,
[n,k,H,W]) per sub-head, averaged.
"""
from __future__ import annotations

from typing import List

import torch
from torch import nn

from contrastyou.arch import UNet
from contrastyou.arch.utils import SingleFeatureExtractor
from contrastyou.hooks.base import EpocherHook, TrainerHook
from contrastyou.meters import AverageValueMeter

decoder_names = UNet.decoder_names
encoder_names = UNet.encoder_names


class DiscreteMITrainHook(TrainerHook):

    @property
    def learnable_modules(self) -> List[nn.Module]:
        return [self._projector]

    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0, num_clusters=20,
                 num_subheads=5, padding=None) -> None:
        super().__init__(hook_name=name)
        assert feature_name in encoder_names + decoder_names, feature_name
        self._feature_name = feature_name
        self._weight = weight
        self._extractor = SingleFeatureExtractor(model, feature_name=feature_name)
        input_dim = model.get_channel_dim(feature_name)
        self._projector = self.init_projector(input_dim=input_dim, num_clusters=num_clusters,
                                              num_subheads=num_subheads)
        self._criterion = self.init_criterion(padding=padding)

    def __call__(self):
        return _DiscreteMIEpochHook(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                                    projector=self._projector, criterion=self._criterion)

    def init_projector(self, *, input_dim, num_clusters, num_subheads=5):
        return self.projector_class(input_dim=input_dim, num_clusters=num_clusters, num_subheads=num_subheads,
                                    head_type="linear", T=1, normalize=False)

    def init_criterion(self, padding: int = None):
        if self._feature_name in encoder_names:
            criterion = self.criterion_class()
            return lambda *args, **kwargs: criterion(*args, **kwargs)[0]
        return self.criterion_class(padding=padding or 0)

    @property
    def projector_class(self):
        from contrastyou.projectors.heads import ClusterHead, DenseClusterHead
        return ClusterHead if self._feature_name in encoder_names else DenseClusterHead

    @property
    def criterion_class(self):
        from contrastyou.losses.discreteMI import IIDLoss, IIDSegmentationLoss
        return IIDLoss if self._feature_name in encoder_names else IIDSegmentationLoss


class _DiscreteMIEpochHook(EpocherHook):

    def __init__(self, *, name: str, weight: float, extractor, projector, criterion) -> None:
        super().__init__(name=name)
        self._extractor = extractor
        self._extractor.bind()
        self._weight = weight
        self._projector = projector
        self._criterion = criterion

    def configure_meters_given_epocher(self, meters):
        meters.register_meter("mi", AverageValueMeter())

    def before_forward_pass(self, **kwargs):
        self._extractor.clear()
        self._extractor.set_enable(True)

    def after_forward_pass(self, **kwargs):
        self._extractor.set_enable(False)

    def _call_implementation(self, *, unlabeled_image, unlabeled_image_tf, affine_transformer, **kwargs):
        n_unl = len(unlabeled_image)
        feature_ = self._extractor.feature()[-n_unl * 2:]
        proj_feature, proj_tf_feature = torch.chunk(feature_, 2, dim=0)
        assert proj_feature.shape == proj_tf_feature.shape
        proj_feature_tf = affine_transformer(proj_feature)
        pairs = [torch.chunk(x, 2, 0) for x in self._projector(torch.cat([proj_feature_tf, proj_tf_feature], dim=0))]
        loss = sum(self._criterion(x1, x2) for x1, x2 in pairs) / len(pairs)
        self.meters["mi"].add(loss.detach())
        return loss * self._weight

    def close(self):
        self._extractor.remove()
