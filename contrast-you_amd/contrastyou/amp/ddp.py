"""rank helpers (contrastyou/amp/ddp.py:8-33)"""
from __future__ import annotations

from typing import Optional

import torch.distributed as dist
from torch import nn


def convert2syncBN(network: nn.Module):
    return nn.SyncBatchNorm.convert_sync_batchnorm(network)


class DDPMixin:
    @property
    def rank(self) -> Optional[int]:
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
        return None

    @property
    def on_master(self) -> bool:
        return self.rank in (0, None)
