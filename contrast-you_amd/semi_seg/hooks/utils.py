"""contrastive label generation (semi_seg/hooks/utils.py:22-102 of the reference)"""
from __future__ import annotations

from semi_seg.epochers.helper import (ACDCCycleGenerator, PartitionLabelGenerator, PatientLabelGenerator,
                                      SIMCLRGenerator)

_GENERATORS = {"partition": PartitionLabelGenerator, "patient": PatientLabelGenerator, "self": SIMCLRGenerator}
_DATASETS = ("acdc", "prostate", "mmwhs", "spleen", "hippocampus")


def global_label_generator(dataset_name: str, contrast_on: str):
    if dataset_name not in _DATASETS:
        raise NotImplementedError(dataset_name)
    if contrast_on == "cycle":
        if dataset_name != "acdc":
            raise NotImplementedError(contrast_on)
        return ACDCCycleGenerator()
    if contrast_on not in _GENERATORS:
        raise NotImplementedError(contrast_on)
    return _GENERATORS[contrast_on]()


def get_label(contrast_on, data_name, partition_group, label_group):
    """list[int] of length n: equal ints = positive pair (same partition / patient / cycle / self)"""
    if data_name == "acdc" or "acdc" in data_name:
        return global_label_generator("acdc", contrast_on)(
            partition_list=partition_group,
            patient_list=[p.split("_")[0] for p in label_group],
            experiment_list=[p.split("_")[1] for p in label_group])
    if data_name in ("prostate", "prostate_md"):
        return global_label_generator("prostate", contrast_on)(
            partition_list=partition_group, patient_list=[p.split("_")[0] for p in label_group])
    if data_name in ("mmwhsct", "mmwhsmr"):
        return global_label_generator("mmwhs", contrast_on)(partition_list=partition_group, patient_list=label_group)
    if data_name in ("spleen", "hippocampus"):
        return global_label_generator(data_name, contrast_on)(partition_list=partition_group,
                                                              patient_list=label_group)
    raise NotImplementedError(data_name)
