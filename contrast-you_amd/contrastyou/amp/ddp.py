"""Process-rank helpers shared by epochers and trainers (the role of contrastyou/amp/ddp.py:8-33
in the reference): logging, meters and checkpoints happen on rank 0 only."""
from __future__ import annotations

from typing import Optional

import torch.distributed as dist
from torch import nn


def _process_rank() -> Optional[int]:
    """rank inside the default process group, None for a single-process run"""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist.get_rank()


class DDPMixin:
    """adds `.rank` / `.on_master` to whatever it is mixed into"""

    rank = property(lambda self: _process_rank())
    on_master = property(lambda self: _process_rank() in (None, 0))


def convert2syncBN(network: nn.Module) -> nn.Module:
    """torch's SyncBatchNorm conversion; the HIP U-Net keeps per-rank statistics (Multi-GPU)"""
    return nn.SyncBatchNorm.convert_sync_batchnorm(network)
