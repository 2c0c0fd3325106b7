"""`get_arch` factory (contrastyou/arch/__init__.py:9-19 of the reference).

`unet` is the hot-path backbone (every named block on the HIP kernels); `unet2` runs its 3x3 conv +
GroupNorm + SiLU stages on the HIP kernels and the attention / strided-conv glue on library ops
(arch/unet2.py); `unetsmp` wraps a third-party ImageNet backbone (a remote weight download,
SURVEY.md section 8c) and is out of scope.
"""
from .unet import UNet, UNetFeatureMapEnum  # noqa: F401
from .unet2 import UNet2  # noqa: F401
from .utils import FeatureExtractor, SingleFeatureExtractor  # noqa: F401

_ARCHS = {"unet": UNet, "unet2": UNet2}


def get_arch(name: str, **kwargs):
    if name == "unetsmp":
        raise NotImplementedError(f"arch `{name}` is outside the HIP hot path of this build")
    assert name in _ARCHS, name
    kwargs.pop("name", None)
    return _ARCHS[name](**kwargs)
