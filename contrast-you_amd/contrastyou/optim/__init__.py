"""Optimizers of the hot path (the reference re-exports torch.optim.*, contrastyou/optim/__init__.py).

`RAdam` here is `FusedRAdam`: torch.optim.RAdam semantics (contrastyou/trainer/base.py:66-75,
config/base.yaml:10-13) as ONE HIP kernel launch per parameter group over flat f32 buffers, with
the data-parallel gradient all-reduce (RCCL) folded into `step()`."""
from torch.optim import SGD, Adam, AdamW  # noqa: F401  (small-parameter fallbacks, e.g. for a discriminator)

from .fused_radam import FlatParams, FusedRAdam  # noqa: F401
from .scheduler import GradualWarmupScheduler  # noqa: F401

RAdam = FusedRAdam
