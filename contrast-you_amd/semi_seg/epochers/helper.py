"""batch unpacking and contrastive label generators (semi_seg/epochers/helper.py:28-71)"""
from __future__ import annotations

from typing import List

from contrastyou.types import to_device


def preprocess_input_with_twice_transformation(data, device, non_blocking=True):
    """{"img":[v1,v2], "gt":[t1,t2], "filename":[f,f], "partition":[p,p], "scan_num":[s,s]}
    -> (v1,t1), (v2,t2), filename, partition, scan"""
    if not isinstance(data["img"], (list, tuple)):
        raise NotImplementedError("twice-transformed batches carry two views per key")
    data = to_device(data, device, non_blocking)
    return (data["img"][0], data["gt"][0]), (data["img"][1], data["gt"][1]), data["filename"][0], \
        data["partition"][0], data["scan_num"][0]


def preprocess_input_with_single_transformation(data, device, non_blocking=True):
    data = to_device(data, device, non_blocking)
    return data["img"], data["gt"], data["filename"], data["partition"], data["scan_num"]


def _encode(values: List[str]) -> List[int]:
    """sklearn.preprocessing.LabelEncoder().fit(v).transform(v): rank among sorted uniques"""
    lut = {v: i for i, v in enumerate(sorted(set(values)))}
    return [lut[v] for v in values]


class PartitionLabelGenerator:
    def __call__(self, partition_list: List[str], **kwargs):
        return _encode(list(partition_list))


class PatientLabelGenerator:
    def __call__(self, patient_list: List[str], **kwargs):
        return _encode(list(patient_list))


class ACDCCycleGenerator:
    def __call__(self, experiment_list: List[str], **kwargs):
        return [0 if e == "00" else 1 for e in experiment_list]


class SIMCLRGenerator:
    def __call__(self, partition_list: List[str], **kwargs):
        return list(range(len(partition_list)))
