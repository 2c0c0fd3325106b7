"""Discrete mutual-information (IIC) regulariser on a tapped feature map -- the hook the reference
defines in semi_seg/hooks/discretemi.py:16-116.

For the two unlabeled views of a batch: tap the features of `feature_name`, warp the plain view's
features with the step's geometry, run both through a cluster head with several sub-heads, and
maximise, per sub-head, the mutual information between the two cluster assignments:

    encoder features  -> ClusterHead       ([n, k] per sub-head) + IIDLoss
    decoder features  -> DenseClusterHead  ([n, k, H, W])        + IIDSegmentationLoss(padding)

The loss is the mean over sub-heads times `weight`.  Heads and losses run on the HIP kernels of
csrc/cy_mi.hip (stacked sub-head projection, grouped softmax, k x k joint, loss on the joint).
"""
from __future__ import annotations

from typing import List

import torch
from torch import nn

from contrastyou.arch import UNet
from contrastyou.arch.utils import SingleFeatureExtractor
from contrastyou.hooks.base import EpocherHook, TrainerHook
from contrastyou.losses.discreteMI import IIDLoss, IIDSegmentationLoss
from contrastyou.meters import AverageValueMeter
from contrastyou.projectors.heads import ClusterHead, DenseClusterHead

decoder_names = UNet.decoder_names
encoder_names = UNet.encoder_names


class _DiscreteMIEpochHook(EpocherHook):

    def __init__(self, *, name: str, weight: float, extractor, projector, criterion) -> None:
        super().__init__(name=name)
        self._weight, self._projector, self._criterion = weight, projector, criterion
        self._extractor = extractor
        self._extractor.bind()

    def configure_meters_given_epocher(self, meters):
        meters.register_meter("mi", AverageValueMeter())

    # the tap records only during the epocher's forward pass
    def before_forward_pass(self, **kwargs):
        self._extractor.clear()
        self._extractor.set_enable(True)

    def after_forward_pass(self, **kwargs):
        self._extractor.set_enable(False)

    def close(self):
        self._extractor.remove()

    def _call_implementation(self, *, unlabeled_image, unlabeled_image_tf, affine_transformer, **kwargs):
        n_unl = len(unlabeled_image)
        plain, transformed = torch.chunk(self._extractor.tail(2 * n_unl), 2, dim=0)
        assert plain.shape == transformed.shape
        both_views = torch.cat([affine_transformer(plain), transformed], dim=0)
        terms = [self._criterion(*torch.chunk(prob, 2, dim=0)) for prob in self._projector(both_views)]
        loss = sum(terms) / len(terms)
        self.meters["mi"].add(loss.detach())
        return loss * self._weight


class DiscreteMITrainHook(TrainerHook):

    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0, num_clusters=20,
                 num_subheads=5, padding=None) -> None:
        super().__init__(hook_name=name)
        assert feature_name in encoder_names + decoder_names, feature_name
        self._feature_name, self._weight = feature_name, weight
        self._extractor = SingleFeatureExtractor(model, feature_name=feature_name)
        self._projector = self.init_projector(input_dim=model.get_channel_dim(feature_name),
                                              num_clusters=num_clusters, num_subheads=num_subheads)
        self._criterion = self.init_criterion(padding=padding)

    @property
    def _on_encoder(self) -> bool:
        return self._feature_name in encoder_names

    @property
    def learnable_modules(self) -> List[nn.Module]:
        return [self._projector]

    # which head / loss pair: pooled vectors for encoder features, dense maps for decoder features
    projector_class = property(lambda self: ClusterHead if self._on_encoder else DenseClusterHead)
    criterion_class = property(lambda self: IIDLoss if self._on_encoder else IIDSegmentationLoss)

    def init_projector(self, *, input_dim, num_clusters, num_subheads=5):
        return self.projector_class(input_dim=input_dim, num_clusters=num_clusters, num_subheads=num_subheads,
                                    head_type="linear", T=1, normalize=False)

    def init_criterion(self, padding: int = None):
        if not self._on_encoder:
            return IIDSegmentationLoss(padding=padding or 0)
        iid = IIDLoss()
        return lambda x1, x2: iid(x1, x2)[0]  # IIDLoss returns (loss, loss without lambda, joint)

    def __call__(self):
        return _DiscreteMIEpochHook(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                                    projector=self._projector, criterion=self._criterion)
