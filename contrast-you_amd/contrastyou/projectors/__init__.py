from .heads import ProjectionHead  # noqa: F401
from .nn import Normalize  # noqa: F401
