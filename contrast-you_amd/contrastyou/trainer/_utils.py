"""small trainer helpers (contrastyou/trainer/_utils.py:41-78)"""
from __future__ import annotations

import contextlib
import os
import tempfile
from pathlib import Path
from typing import Union

import torch


def safe_save(checkpoint_dictionary, save_path):
    """write to a temp file in the target directory, then rename: a crash never leaves a torn file"""
    save_path = Path(save_path)
    fd, tmp = tempfile.mkstemp(prefix="tmp_ckpt_", suffix=".pth", dir=str(save_path.parent))
    try:
        with os.fdopen(fd, "wb") as f:
            torch.save(checkpoint_dictionary, f)
        os.replace(tmp, str(save_path))
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


@contextlib.contextmanager
def create_save_dir(self, save_dir: Union[Path, str]):
    """relative `save_dir` -> RUN_PATH/save_dir; absolute stays; created on exit"""
    save_dir = str(save_dir)
    if not Path(save_dir).is_absolute():
        save_dir = str(Path(self.RUN_PATH) / save_dir)
    yield save_dir
    Path(save_dir).mkdir(exist_ok=True, parents=True)


def run_once(f):
    def wrapper(*args, **kwargs):
        if wrapper.has_run:
            raise RuntimeError(f"{f} has been called more than once.")
        wrapper.has_run = True
        return f(*args, **kwargs)

    wrapper.has_run = False
    return wrapper
