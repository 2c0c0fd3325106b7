"""cyhip: Python face of libcontrastyou_hip.so (hand-written gfx950 kernels).

`ops` = one function per kernel family over torch device tensors; `functions` = the
torch.autograd.Function seam used by the reference-compatible modules in `contrastyou`.
"""
from . import _lib  # noqa: F401
from ._lib import HipExtensionMissing, HipKernelError, load  # noqa: F401
