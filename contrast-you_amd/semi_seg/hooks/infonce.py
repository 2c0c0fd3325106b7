"""InfoNCE regulariser on an encoder feature map (semi_seg/hooks/infonce.py:84-245).

`INFONCEHook` (TrainerHook) owns the feature tap, the projection head and the SupCon criterion;
once per epoch it hands out an `_INFONCEEpochHook` whose `_call_implementation`:
    features of the last 2*n_unl slices of the forward pass -> (unlabeled, unlabeled_tf)
    -> affine(features of the untransformed view) -> projector(cat) -> labels -> SupConLoss1
exactly in the reference's order.  The loss value is metered on the device; the criterion's
unit-norm / NaN checks are deferred to `close()` (once per epoch) instead of syncing per batch.
"""
from __future__ import annotations

import typing as t
from functools import partial
from typing import List, Union

import torch
from torch import nn

from contrastyou.arch.utils import SingleFeatureExtractor
from contrastyou.hooks.base import EpocherHook, TrainerHook
from contrastyou.losses.contrastive import SupConLoss1
from contrastyou.meters import AverageValueMeter, MeterInterface
from cyhip import parallel

from .utils import get_label


class INFONCEHook(TrainerHook):

    @property
    def learnable_modules(self) -> List[nn.Module]:
        return [self._projector]

    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0,
                 spatial_size: t.Sequence[int] = None, data_name: str, contrast_on: str,
                 global_negatives: bool = False) -> None:
        super().__init__(hook_name=name)
        # data-parallel runs: contrast against the embeddings of every rank (all-gather over RCCL)
        self._global_negatives = global_negatives
        self.register_non_trackable_buffer("_model", model)
        assert feature_name in model.arch_elements, feature_name
        self._feature_name = feature_name
        self._weight = weight
        self._extractor = SingleFeatureExtractor(model, feature_name=feature_name)
        input_dim = model.get_channel_dim(feature_name)
        if self.is_encoder:
            assert (spatial_size is None) or (tuple(spatial_size) == (1, 1)), spatial_size
            spatial_size = (1, 1)
        else:
            raise NotImplementedError("dense (decoder-feature) InfoNCE is the next scope row of this build "
                                      "(SURVEY.md section 8f); encoder features only")
        self._projector = self.init_projector(input_dim=input_dim, spatial_size=spatial_size)
        self._criterion = self.init_criterion()
        self._label_generator = partial(get_label, contrast_on=contrast_on, data_name=data_name)

    def __call__(self):
        return _INFONCEEpochHook(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                                 projector=self._projector, criterion=self._criterion,
                                 label_generator=self._label_generator,
                                 global_negatives=self._global_negatives)

    def init_criterion(self) -> SupConLoss1:
        self._criterion = SupConLoss1()
        return self._criterion

    def init_projector(self, *, input_dim, spatial_size):
        return self.projector_class(input_dim=input_dim, hidden_dim=256, output_dim=256, head_type="mlp",
                                    normalize=True, spatial_size=spatial_size)

    @property
    def projector_class(self):
        from contrastyou.projectors.heads import ProjectionHead
        return ProjectionHead

    @property
    def is_encoder(self):
        return self._feature_name in self._model.encoder_names


class _INFONCEEpochHook(EpocherHook):

    def __init__(self, *, name: str, weight: float, extractor, projector, criterion: Union[SupConLoss1],
                 label_generator, global_negatives: bool = False) -> None:
        super().__init__(name=name)
        self._global_negatives = global_negatives
        self._extractor = extractor
        self._extractor.bind()
        self._weight = weight
        self._projector = projector
        self._criterion = criterion
        self._criterion.defer_checks = True
        self._label_generator = label_generator
        self._n = 0

    def configure_meters_given_epocher(self, meters: MeterInterface):
        meters = super().configure_meters_given_epocher(meters)
        meters.register_meter("loss", AverageValueMeter())
        return meters

    def before_forward_pass(self, **kwargs):
        self._extractor.clear()
        self._extractor.set_enable(True)

    def after_forward_pass(self, **kwargs):
        self._extractor.set_enable(False)

    def _call_implementation(self, *, affine_transformer, seed, unlabeled_tf_logits, unlabeled_logits_tf,
                             partition_group, label_group, **kwargs):
        n_unl = len(unlabeled_logits_tf)
        feature_ = self._extractor.feature()[-n_unl * 2:]
        unlabeled_features, unlabeled_tf_features = torch.chunk(feature_, 2, dim=0)
        unlabeled_features_tf = affine_transformer(unlabeled_features)
        norm_features_tf, norm_tf_features = torch.chunk(
            self._projector(torch.cat([unlabeled_features_tf, unlabeled_tf_features], dim=0)), 2)
        if self._global_negatives and parallel.is_parallel():
            # same label semantics on the concatenated groups of all ranks; every rank evaluates the
            # full matrix, autograd follows its own rows, and the loss is scaled by world_size
            # because the optimizer averages gradients over ranks
            labels = self._label_generator(partition_group=parallel.gather_labels(partition_group),
                                           label_group=parallel.gather_labels(label_group))
            loss = self._criterion(parallel.gather_cat(norm_features_tf), parallel.gather_cat(norm_tf_features),
                                   target=labels) * parallel.world_size()
        else:
            labels = self._label_generator(partition_group=partition_group, label_group=label_group)
            loss = self._criterion(norm_features_tf, norm_tf_features, target=labels)
        self.meters["loss"].add(loss.detach())
        self._n += 1
        return loss * self._weight

    def close(self):
        self._extractor.remove()
        self._criterion.validate()  # the reference's per-batch asserts, once per epoch
        self._criterion.defer_checks = False
