"""`get_arch` factory (contrastyou/arch/__init__.py:9-19 of the reference).

Only `unet` is built on the HIP path this package implements; `unet2` / `unetsmp` are other
backbones outside the SemiSupervisedEpocher+InfoNCE hot path.
"""
from .unet import UNet, UNetFeatureMapEnum  # noqa: F401
from .utils import FeatureExtractor, SingleFeatureExtractor  # noqa: F401

_ARCHS = {"unet": UNet}


def get_arch(name: str, **kwargs):
    if name in ("unet2", "unetsmp"):
        raise NotImplementedError(f"arch `{name}` is outside the HIP hot path of this build")
    assert name in _ARCHS, name
    kwargs.pop("name", None)
    return _ARCHS[name](**kwargs)
