"""hook factories (semi_seg/hooks/creator.py:31-49,51-52,92-122 of the reference)"""
from __future__ import annotations

from typing import List, Sequence, Union

from torch import nn

from contrastyou.arch._base import sort_arch
from contrastyou.hooks.base import CombineTrainerHook, TrainerHook
from contrastyou.utils.utils import ntuple

from .consistency import ConsistencyTrainerHook
from .infonce import INFONCEHook
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
