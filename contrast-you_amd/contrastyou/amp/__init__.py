from .amp import AMPScaler, BF16Scaler  # noqa: F401
from .ddp import DDPMixin, convert2syncBN  # noqa: F401
