"""Data-parallel plumbing over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X).

One process per GPU.  What crosses ranks on the hot path:
  * gradients  -- ONE all-reduce(AVG) per flat optimizer buffer inside FusedRAdam.step()
                  (contrastyou/optim/fused_radam.py): 35 MB f32 for the U-Net + 0.8 MB projector;
  * embeddings -- `gather_cat` for InfoNCE with global negatives (BASELINE config 5): every rank
                  evaluates the loss on the full similarity matrix; autograd only follows this
                  rank's rows, and the loss is scaled by world_size so that the gradient mean the
                  optimizer takes over ranks equals the gradient of the global loss.
BatchNorm statistics stay per rank (what DistributedDataParallel does to plain BatchNorm2d; the
reference never converts to SyncBN, contrastyou/amp/ddp.py:8-9).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


def is_parallel() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank() -> int:
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend: str = "nccl") -> int:
    """torchrun-style init (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); returns the local rank"""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return local_rank


def gather_cat(z: Tensor) -> Tensor:
    """[n, D] on every rank -> [world*n, D] in rank order; differentiable w.r.t. the local rows"""
    if not is_parallel():
        return z
    parts: List[Tensor] = [torch.empty_like(z) for _ in range(world_size())]
    dist.all_gather(parts, z.detach().contiguous())
    parts[rank()] = z
    return torch.cat(parts, dim=0)


def gather_labels(labels: Sequence) -> list:
    """concatenate per-rank label lists (arbitrary hashables) in rank order"""
    if not is_parallel():
        return list(labels)
    out: List[list] = [None] * world_size()  # type: ignore
    dist.all_gather_object(out, list(labels))
    return [v for part in out for v in part]
