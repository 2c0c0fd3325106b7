from ._utils import create_save_dir, safe_save  # noqa: F401
from .base import Trainer  # noqa: F401
