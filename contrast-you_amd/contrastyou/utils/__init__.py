from .general import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
