"""InfoNCE regulariser on an encoder feature map (semi_seg/hooks/infonce.py:84-245), its dense
variant on decoder feature maps (`_INFONCEDenseHook`, infonce.py:251-279; `region_extractor`,
infonce.py:31-46), the self-paced variant (`SelfPacedINFONCEHook` + `PScheduler`, infonce.py:58-80,146-178,
281-305) and the superpixel-labelled dense variant (`SuperPixelInfoNCEHook`, infonce.py:180-194,308-340).

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

import numpy as np
import torch
from torch import nn

from contrastyou.arch.utils import SingleFeatureExtractor
from contrastyou.hooks.base import EpocherHook, TrainerHook
from contrastyou.losses.contrastive import SelfPacedSupConLoss, SupConLoss1
from contrastyou.meters import AverageValueMeter, MeterInterface
from contrastyou.utils.utils import fix_all_seed_for_transforms
from cyhip import ops, parallel
from cyhip.functions import GatherRowsFn

from .utils import get_label


def region_points(n: int, h: int, w: int, *, point_nums: int = 5, seed: int):
    """the (row, col) samples of `region_extractor` for n maps of h x w: under the seeded RNGs, per
    image `point_nums` distinct rows then `point_nums` distinct columns, zipped (infonce.py:39-46)"""
    with fix_all_seed_for_transforms(seed):
        return [list(zip((int(a) for a in np.random.choice(range(h), point_nums, replace=False)),
                         (int(b) for b in np.random.choice(range(w), point_nums, replace=False))))
                for _ in range(n)]


def region_extractor(normalize_features: torch.Tensor, *, point_nums=5, seed: int) -> torch.Tensor:
    """[n,D,h,w] -> [point_nums*n, D]: the sampled feature vectors, image-major (infonce.py:31-46);
    one HIP row gather instead of 5n indexing kernels"""
    n, d, h, w = normalize_features.shape
    pts = region_points(n, h, w, point_nums=point_nums, seed=seed)
    idx = np.asarray([(i * h + a) * w + b for i, im in enumerate(pts) for a, b in im], dtype=np.int32)
    rows = normalize_features.permute(0, 2, 3, 1).reshape(n * h * w, d)
    return GatherRowsFn.apply(rows, ops.pinned.upload(torch.from_numpy(idx), normalize_features.device))


class PScheduler:
    """gamma of the self-paced loss over the epochs (infonce.py:58-80):
    begin + (end - begin) * (epoch / max_epoch) ** p"""

    def __init__(self, max_epoch, begin_value=0.0, end_value=1.0, p=0.5):
        self.max_epoch = max_epoch
        self.begin_value = float(begin_value)
        self.end_value = float(end_value)
        self.epoch = 0
        self.p = p

    def step(self):
        self.epoch += 1

    @property
    def value(self):
        return self.get_lr(self.epoch)

    def get_lr(self, cur_epoch):
        return self.begin_value + (self.end_value - self.begin_value) * np.power(cur_epoch / self.max_epoch, self.p)


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
            assert isinstance(spatial_size, t.Sequence) and isinstance(tuple(spatial_size)[0], int), spatial_size
        self._projector = self.init_projector(input_dim=input_dim, spatial_size=spatial_size)
        self._criterion = self.init_criterion()
        self._label_generator = partial(get_label, contrast_on=contrast_on, data_name=data_name)

    def __call__(self):
        cls = _INFONCEEpochHook if self.is_encoder else _INFONCEDenseHook
        return cls(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                   projector=self._projector, criterion=self._criterion, label_generator=self._label_generator,
                   global_negatives=self._global_negatives)

    def init_criterion(self) -> SupConLoss1:
        self._criterion = SupConLoss1()
        return self._criterion

    def init_projector(self, *, input_dim, spatial_size):
        return self.projector_class(input_dim=input_dim, hidden_dim=256, output_dim=256, head_type="mlp",
                                    normalize=True, spatial_size=spatial_size)

    @property
    def projector_class(self):
        from contrastyou.projectors.heads import DenseProjectionHead, ProjectionHead
        return ProjectionHead if self.is_encoder else DenseProjectionHead

    @property
    def is_encoder(self):
        return self._feature_name in self._model.encoder_names


class SelfPacedINFONCEHook(INFONCEHook):
    """self-paced contrastive loss per layer (infonce.py:146-178): every epoch's hook gets the criterion with the
    scheduler's current gamma, then the scheduler steps"""

    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0, spatial_size=(1, 1),
                 data_name: str, contrast_on: str, mode="soft", p=0.5, begin_value=1e6, end_value=1e6,
                 correct_grad: bool = False, max_epoch: int) -> None:
        self._mode = mode
        self._p = float(p)
        self._begin_value = float(begin_value)
        self._end_value = float(end_value)
        self._max_epoch = int(max_epoch)
        self._correct_grad = correct_grad
        super().__init__(name=name, model=model, feature_name=feature_name, weight=weight, spatial_size=spatial_size,
                         data_name=data_name, contrast_on=contrast_on)

    def init_criterion(self) -> SelfPacedSupConLoss:
        self._scheduler = PScheduler(max_epoch=self._max_epoch, begin_value=self._begin_value,
                                     end_value=self._end_value, p=self._p)
        self._criterion = SelfPacedSupConLoss(weight_update=self._mode, correct_grad=self._correct_grad)
        return self._criterion

    def __call__(self):
        gamma = self._scheduler.value
        self._scheduler.step()
        self._criterion.set_gamma(gamma)
        return _SPINFONCEEpochHook(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                                   projector=self._projector, criterion=self._criterion,
                                   label_generator=self._label_generator)


class SuperPixelInfoNCEHook(INFONCEHook):
    """dense InfoNCE whose classes are the superpixel ids under the sampled positions (infonce.py:180-194);
    decoder features only"""

    def __init__(self, *, name, model: nn.Module, feature_name: str, weight: float = 1.0,
                 spatial_size: t.Sequence[int] = None, data_name: str, contrast_on: str) -> None:
        super().__init__(name=name, model=model, feature_name=feature_name, weight=weight, spatial_size=spatial_size,
                         data_name=data_name, contrast_on=contrast_on)
        assert self.is_encoder is False, f"{self.__class__.__name__} only supports decoder features"

    def __call__(self) -> "_SuperPixelInfoNCEEPochHook":
        return _SuperPixelInfoNCEEPochHook(name=self._hook_name, weight=self._weight, extractor=self._extractor,
                                           projector=self._projector, criterion=self._criterion,
                                           label_generator=self._label_generator)


class _INFONCEEpochHook(EpocherHook):

    def __init__(self, *, name: str, weight: float, extractor, projector,
                 criterion: Union[SupConLoss1, SelfPacedSupConLoss], label_generator,
                 global_negatives: bool = False) -> None:
        super().__init__(name=name)
        self._global_negatives = global_negatives
        self._extractor = extractor
        self._extractor.bind()
        self._weight = weight
        self._projector = projector
        self._criterion = criterion
        self._criterion.defer_checks = True  # (SupConLoss1: checks once per epoch; a no-op attribute elsewhere)
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
        feature_ = self._extractor.tail(n_unl * 2)
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
        if hasattr(self._criterion, "validate"):
            self._criterion.validate()  # the reference's per-batch asserts, once per epoch
        self._criterion.defer_checks = False


class _SPINFONCEEpochHook(_INFONCEEpochHook):
    """`_INFONCEEpochHook` + the self-paced meters (infonce.py:281-305)"""
    _criterion: SelfPacedSupConLoss

    def configure_meters_given_epocher(self, meters: MeterInterface):
        meters = super().configure_meters_given_epocher(meters)
        meters.register_meter("sp_weight", AverageValueMeter())
        meters.register_meter("age_param", AverageValueMeter())
        return meters

    def _call_implementation(self, **kwargs):
        loss = super()._call_implementation(**kwargs)
        self.meters["sp_weight"].add(self._criterion.downgrade_ratio)
        self.meters["age_param"].add(self._criterion.age_param)
        return loss


def nearest_source_index(dst: int, in_size: int, out_size: int) -> int:
    """source index of F.interpolate(mode="nearest") for output index `dst`: min(floor(dst * float32(in / out)), in - 1)
    with the scale and the product in float32, as ATen computes it -- integer (dst * in) // out differs from that for
    some sizes (224 -> 48: dst 21 reads 97, not 98)"""
    import numpy as np
    scale = np.float32(in_size) / np.float32(out_size)
    return min(int(np.floor(np.float32(dst) * scale)), in_size - 1)


class _SuperPixelInfoNCEEPochHook(_INFONCEEpochHook):
    """infonce.py:308-340: project both views densely, sample 5 positions per image (same seed for both views and
    for the superpixel map), label every sampled vector with the superpixel id under it"""

    def _call_implementation(self, *, affine_transformer, seed, unlabeled_tf_logits, unlabeled_logits_tf,
                             partition_group, label_group, batch_data: t.Dict[str, t.Any] = None, **kwargs):
        assert batch_data is not None
        n_unl = len(unlabeled_logits_tf)
        feature_ = self._extractor.tail(n_unl * 2)
        unlabeled_features, unlabeled_tf_features = torch.chunk(feature_, 2, dim=0)
        unlabeled_features_tf = affine_transformer(unlabeled_features, seed=seed)
        sh, sw = self._projector._spatial_size
        pts = region_points(n_unl, sh, sw, point_nums=5, seed=seed)
        rows = self._projector.project_points(torch.cat([unlabeled_features_tf, unlabeled_tf_features], dim=0),
                                              pts + pts)
        norm_features_tf_selected, norm_tf_features_selected = torch.chunk(rows, 2)
        dev = rows.device
        superpixel_mask = (batch_data["superpixel"][0].to(dev) * 255.0).type(torch.uint8).float()
        superpixel_mask_tf = affine_transformer(superpixel_mask)
        # F.interpolate(mode="nearest") to (sh, sw), then the sampled positions: the source index of ATen's nearest
        # kernel (float32 scale arithmetic, `nearest_source_index`), all positions with ONE gather and one host read
        H, W = superpixel_mask_tf.shape[-2:]
        ii = torch.tensor([i for i, im in enumerate(pts) for _ in im], device=dev)
        aa = torch.tensor([nearest_source_index(a, H, sh) for im in pts for a, _ in im], device=dev)
        bb = torch.tensor([nearest_source_index(b, W, sw) for im in pts for _, b in im], device=dev)
        labels = [int(v) & 0xff for v in superpixel_mask_tf[ii, 0, aa, bb].tolist()]  # (.type(torch.uint8), infonce.py:336)
        loss = self._criterion(norm_features_tf_selected, norm_tf_features_selected, target=labels)
        self.meters["loss"].add(loss.detach())
        self._n += 1
        return loss * self._weight


class _INFONCEDenseHook(_INFONCEEpochHook):
    """InfoNCE between 5 sampled positions per image of the pooled, projected decoder feature maps
    of the two views; every sampled vector is its own class (infonce.py:251-279).

    The reference projects the full [2n,256,s,s] maps and then keeps 5 of the s*s positions; here
    `DenseProjectionHead.project_points` evaluates exactly those bins (identical values -- the
    samples depend on (seed, n, s) only), so the 1x1-conv MLP runs on ~1% of the pixels."""

    def _call_implementation(self, *, affine_transformer, seed, unlabeled_tf_logits, unlabeled_logits_tf,
                             partition_group, label_group, **kwargs):
        n_unl = len(unlabeled_logits_tf)
        feature_ = self._extractor.tail(n_unl * 2)
        unlabeled_features, unlabeled_tf_features = torch.chunk(feature_, 2, dim=0)
        unlabeled_features_tf = affine_transformer(unlabeled_features, seed=seed)
        sh, sw = self._projector._spatial_size
        pts = region_points(n_unl, sh, sw, point_nums=5, seed=seed)  # same draw for both views (same seed)
        rows = self._projector.project_points(torch.cat([unlabeled_features_tf, unlabeled_tf_features], dim=0),
                                              pts + pts)
        norm_features_tf_selected, norm_tf_features_selected = torch.chunk(rows, 2)
        labels = list(range(norm_features_tf_selected.shape[0]))
        loss = self._criterion(norm_features_tf_selected, norm_tf_features_selected, target=labels)
        self.meters["loss"].add(loss.detach())
        self._n += 1
        return loss * self._weight
