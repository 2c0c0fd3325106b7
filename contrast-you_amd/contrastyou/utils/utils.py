"""host-side helpers used inside the step (subset of contrastyou/utils/utils.py)"""
from __future__ import annotations

import os
import random
from contextlib import contextmanager
from typing import List

import numpy as np
import torch
from torch import nn

__all__ = ["fix_all_seed", "fix_all_seed_for_transforms", "fix_all_seed_within_context", "class_name",
           "get_lrs_from_optimizer", "disable_tracking_bn_stats", "get_model", "get_dataset",
           "average_iter", "ntuple", "ignore_exception", "extract_model_state_dict"]


def fix_all_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


@contextmanager
def fix_all_seed_for_transforms(seed):
    """seed python / numpy / torch-CPU RNGs inside the block, restore afterwards (utils.py:131)"""
    st = (random.getstate(), np.random.get_state(), torch.random.get_rng_state())
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    try:
        yield
    finally:
        random.setstate(st[0])
        np.random.set_state(st[1])
        torch.random.set_rng_state(st[2])


@contextmanager
def fix_all_seed_within_context(seed):
    st = (random.getstate(), np.random.get_state(), torch.random.get_rng_state())
    cuda = torch.cuda.is_available()
    cst = torch.cuda.get_rng_state_all() if cuda else None
    fix_all_seed(seed)
    try:
        yield
    finally:
        random.setstate(st[0])
        np.random.set_state(st[1])
        torch.random.set_rng_state(st[2])
        if cuda:
            torch.cuda.set_rng_state_all(cst)


def class_name(obj) -> str:
    return obj.__class__.__name__


def get_lrs_from_optimizer(optimizer) -> List[float]:
    return [g["lr"] for g in optimizer.param_groups]


@contextmanager
def disable_tracking_bn_stats(model: nn.Module):
    """flip track_running_stats of every BN inside the block (utils.py:225-237): batch statistics
    are used, running statistics are left untouched"""

    def flip(m):
        if hasattr(m, "track_running_stats"):
            m.track_running_stats ^= True

    model.apply(flip)
    try:
        yield
    finally:
        model.apply(flip)


def get_model(model):
    if isinstance(model, (nn.parallel.DistributedDataParallel, nn.parallel.DataParallel)):
        return model.module
    if hasattr(model, "module") and isinstance(getattr(model, "module"), nn.Module) and \
            model.__class__.__name__ == "DataParallelStep":
        return model.module
    if isinstance(model, nn.Module):
        return model
    raise TypeError(type(model))


def get_dataset(dataloader):
    if hasattr(dataloader, "dataset"):
        return dataloader.dataset
    if hasattr(dataloader, "_dataset"):
        return dataloader._dataset
    raise AttributeError(f"cannot find a dataset on {type(dataloader)}")


def average_iter(values):
    values = list(values)
    return sum(values) / len(values)


def ntuple(n):
    def parse(x):
        if isinstance(x, str):
            return (x,) * n
        try:
            x = list(x)
        except TypeError:
            return (x,) * n
        if len(x) == 1:
            return (x[0],) * n
        if len(x) != n:
            raise RuntimeError(f"inconsistent shape between {x} and {n}")
        return tuple(x)

    return parse


@contextmanager
def ignore_exception(*exc):
    try:
        yield
    except (exc or (Exception,)):
        pass


def extract_model_state_dict(trainer_checkpoint_path: str, *, keyword="_model"):
    """model weights out of a trainer checkpoint (`last.pth` / `best.pth`): the `_model.*` entries of
    its `module_state` section (contrastyou/utils/utils.py:88-90).  Loaded with `weights_only=True`:
    nothing in the file is executed; a checkpoint that pickles custom objects is refused by torch."""
    state = torch.load(trainer_checkpoint_path, map_location="cpu", weights_only=True)
    prefix = keyword + "."
    return {k[len(prefix):]: v for k, v in state["module_state"].items() if k.startswith(prefix)}
