"""hook factories (semi_seg/hooks/creator.py:31-49,51-52,92-122 of the reference)"""
from __future__ import annotations

from typing import List, Sequence, Union

from torch import nn

from contrastyou.arch._base import sort_arch
from contrastyou.hooks.base import CombineTrainerHook, TrainerHook
from contrastyou.utils.utils import ntuple

from .consistency import ConsistencyTrainerHook
from .discretemi import DiscreteMITrainHook, decoder_names
from .infonce import INFONCEHook, SelfPacedINFONCEHook, SuperPixelInfoNCEHook
from .midl import IIDSegmentationTrainerHook
from .mt import MeanTeacherTrainerHook


def get_individual_hook(*hooks):
    for h in hooks:
        assert isinstance(h, TrainerHook)
        if isinstance(h, CombineTrainerHook):
            yield from get_individual_hook(*h._hooks)  # noqa
        else:
            yield h


def mt_in_hooks(*hooks) -> bool:
    return any(isinstance(h, MeanTeacherTrainerHook) for h in get_individual_hook(*hooks))


def feature_until_from_hooks(*hooks, model) -> Union[str, None]:
    names = [h._feature_name for h in get_individual_hook(*hooks) if hasattr(h, "_feature_name")]
    if names:
        return sort_arch(names, model=model)[-1]
    return model.arch_elements[-1]


def create_consistency_hook(weight: float):
    return ConsistencyTrainerHook(name="consistency", weight=weight)


def create_mt_hook(*, model: nn.Module, weight: float, alpha: float = 0.999, weight_decay: float = 1e-5,
                   update_bn: bool = False, hard_clip: bool = False):
    return MeanTeacherTrainerHook(name="mt", model=model, weight=weight, alpha=alpha, weight_decay=weight_decay,
                                  update_bn=update_bn, hard_clip=hard_clip)


def _infonce_hook(*, model: nn.Module, feature_name: str, weight: float, contrast_on: str, data_name: str,
                  spatial_size: int, global_negatives: bool = False):
    return INFONCEHook(name=f"infonce/{feature_name}/{contrast_on}", model=model, feature_name=feature_name,
                       weight=weight, data_name=data_name, contrast_on=contrast_on,
                       spatial_size=(spatial_size, spatial_size), global_negatives=global_negatives)


def create_infonce_hooks(*, model: nn.Module, feature_names: Union[str, List[str]],
                         weights: Union[float, List[float]], contrast_ons: Union[str, List[str]],
                         spatial_size: Union[int, Sequence[int]] = 1, data_name: str,
                         global_negatives: bool = False):
    n = 1 if isinstance(feature_names, str) else len(feature_names)
    rep = ntuple(n)
    hooks = [_infonce_hook(model=model, feature_name=f, weight=w, contrast_on=c, data_name=data_name,
                           spatial_size=ss, global_negatives=global_negatives)
             for f, w, c, ss in zip(rep(feature_names), rep(weights), rep(contrast_ons), rep(spatial_size))]
    return CombineTrainerHook(*hooks)


def _infonce_sp_hook(*, model: nn.Module, feature_name: str, weight: float, contrast_on: str, data_name: str,
                     begin_value: float = 1e6, end_value: float = 1e6, max_epoch: int, mode: str = "soft", p=0.5,
                     correct_grad=False):
    return SelfPacedINFONCEHook(name=f"spinfoce/{feature_name}/{contrast_on}", model=model, feature_name=feature_name,
                                weight=weight, data_name=data_name, contrast_on=contrast_on, mode=mode, p=p,
                                begin_value=begin_value, end_value=end_value, max_epoch=max_epoch,
                                correct_grad=correct_grad)


def create_sp_infonce_hooks(*, model: nn.Module, feature_names: Union[str, List[str]],
                            weights: Union[float, List[float]], contrast_ons: Union[str, List[str]], data_name: str,
                            begin_values: Union[float, List[float]] = 1e10,
                            end_values: Union[float, List[float]] = 1e10, mode: str, p=0.5, max_epoch: int,
                            correct_grad: Union[bool, List[bool]] = False):
    """one SelfPacedINFONCEHook per feature (creator.py:122-145)"""
    n = 1 if isinstance(feature_names, str) else len(feature_names)
    rep = ntuple(n)
    hooks = [_infonce_sp_hook(model=model, feature_name=f, weight=w, contrast_on=c, data_name=data_name,
                              begin_value=b, end_value=e, max_epoch=max_epoch, mode=mode, p=p, correct_grad=g)
             for f, w, c, b, e, g in zip(rep(feature_names), rep(weights), rep(contrast_ons), rep(begin_values),
                                         rep(end_values), rep(correct_grad))]
    return CombineTrainerHook(*hooks)


def _create_superpixel_hook(*, model: nn.Module, feature_name: str, weight: float, data_name: str,
                            spatial_size: int):
    return SuperPixelInfoNCEHook(name=f"infonce/{feature_name}/superpixel", model=model, feature_name=feature_name,
                                 weight=weight, data_name=data_name, contrast_on="self",
                                 spatial_size=(spatial_size, spatial_size))


def create_superpixel_hooks(*, model: nn.Module, feature_names: Union[str, List[str]],
                            weights: Union[float, List[float]], spatial_size: Union[int, Sequence[int]],
                            data_name: str):
    """creator.py:263-281"""
    n = 1 if isinstance(feature_names, str) else len(feature_names)
    rep = ntuple(n)
    hooks = [_create_superpixel_hook(model=model, feature_name=f, weight=w, data_name=data_name, spatial_size=ss)
             for f, w, ss in zip(rep(feature_names), rep(weights), rep(spatial_size))]
    return CombineTrainerHook(*hooks)


def create_discrete_mi_hooks(*, feature_names: List[str], weights: List[float], paddings: List[int],
                             model: nn.Module):
    """one DiscreteMITrainHook per feature; `paddings` lists the displacement radius of the DECODER
    features only, in order (creator.py:56-67)"""
    assert len(feature_names) == len(weights), (feature_names, weights)
    decoder_features = [f for f in feature_names if f in decoder_names]
    assert len(paddings) == len(decoder_features), (decoder_features, paddings)
    pad_gen = iter(paddings)
    paddings_ = [next(pad_gen) if f in decoder_features else None for f in feature_names]
    hooks = [DiscreteMITrainHook(name=f"discreteMI/{f.lower()}", model=model, feature_name=f, weight=w, padding=p)
             for f, w, p in zip(feature_names, weights, paddings_)]
    return CombineTrainerHook(*hooks)


def create_discrete_mi_consistency_hook(*, model: nn.Module, feature_names: Union[str, List[str]],
                                        mi_weights: Union[float, List[float]], dense_paddings: List[int] = None,
                                        consistency_weight: float):
    """config/hooks/udaiic.yaml: IIC on feature maps + output consistency (creator.py:76-90)"""
    n = 1 if isinstance(feature_names, str) else len(feature_names)
    feature_names = ntuple(n)(feature_names)
    mi_weights = ntuple(n)(mi_weights)
    n_dense = len([f for f in feature_names if f in decoder_names])
    dense_paddings = ntuple(n_dense)(dense_paddings) if n_dense else ()
    mi = create_discrete_mi_hooks(feature_names=list(feature_names), weights=list(mi_weights),
                                  paddings=list(dense_paddings), model=model)
    return CombineTrainerHook(mi, create_consistency_hook(weight=consistency_weight))


def create_iid_segmentation_hook(*, weight: float, mi_lambda: float = 1.0):
    """config/hooks/iid.yaml: IIC between the two views' segmentation outputs"""
    return IIDSegmentationTrainerHook(hook_name="midl_hook", weight=weight, mi_lambda=mi_lambda)
