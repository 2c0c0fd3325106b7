"""mixed-precision step helpers (AMPScaler, the bf16 pass-through scaler) and rank helpers"""
from .amp import AMPScaler, BF16Scaler  # noqa: F401
from .ddp import DDPMixin, convert2syncBN  # noqa: F401
