from .contrastive import SelfPacedSupConLoss, SupConLoss1, is_normalized  # noqa: F401
from .redundancy_reduction import RedundancyCriterion  # noqa: F401
from .discreteMI import IIDLoss, IIDSegmentationLoss  # noqa: F401
from .kl import KL_div, Entropy  # noqa: F401
