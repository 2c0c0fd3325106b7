from .heads import ClusterHead, DenseClusterHead, DenseProjectionHead, ProjectionHead  # noqa: F401
from .nn import Normalize  # noqa: F401
