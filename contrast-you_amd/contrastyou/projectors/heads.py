"""Projection head of the encoder InfoNCE hook (contrastyou/projectors/heads.py:12-22,81-96):
AdaptiveAvgPool2d(1) -> Flatten -> Linear -> LeakyReLU(0.01) -> Linear -> L2 normalise.

`_header` is an nn.Sequential with the reference's layout, so parameter names
(`_header.2.weight`, `_header.4.bias`, ...) and checkpoints interchange; its children only
hold parameters -- forward runs the HIP kernels (cyhip.functions.AvgPoolFn / LinearFn /
L2NormFn)."""
from __future__ import annotations

from torch import Tensor, nn

from cyhip.functions import AvgPoolFn, L2NormFn, LinearFn

from .nn import Flatten, Identical, Normalize

__all__ = ["ProjectionHead"]


class ProjectionHead(nn.Module):

    def __init__(self, *, input_dim: int, hidden_dim=256, output_dim: int, head_type: str, normalize: bool,
                 pool_name="adaptive_avg", spatial_size=(1, 1)):
        super().__init__()
        assert head_type in ("mlp", "linear"), head_type
        assert pool_name in ("adaptive_avg", "adaptive_max"), pool_name
        if pool_name != "adaptive_avg" or tuple(spatial_size) != (1, 1):
            raise NotImplementedError("the HIP projection head implements adaptive_avg pooling to (1, 1), "
                                      "the only configuration INFONCEHook creates for encoder features")
        self._input_dim, self._output_dim = input_dim, output_dim
        self._head_type, self._normalize = head_type, normalize
        pool = nn.AdaptiveAvgPool2d((1, 1))
        tail = Normalize() if normalize else Identical()
        if head_type == "mlp":
            self._header = nn.Sequential(pool, Flatten(), nn.Linear(input_dim, hidden_dim),
                                         nn.LeakyReLU(0.01, inplace=True), nn.Linear(hidden_dim, output_dim), tail)
        else:
            self._header = nn.Sequential(pool, Flatten(), nn.Linear(input_dim, output_dim), tail)

    def forward(self, features: Tensor) -> Tensor:
        h = self._header
        x = AvgPoolFn.apply(features)
        if self._head_type == "mlp":
            x = LinearFn.apply(x, h[2].weight, h[2].bias, 1, 0.01)
            x = LinearFn.apply(x, h[4].weight, h[4].bias, 0, 0.0)
        else:
            x = LinearFn.apply(x, h[2].weight, h[2].bias, 0, 0.0)
        return L2NormFn.apply(x) if self._normalize else x
