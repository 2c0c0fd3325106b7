"""Projection head of the encoder InfoNCE hook (contrastyou/projectors/heads.py:12-22,81-96):
AdaptiveAvgPool2d(1) -> Flatten -> Linear -> LeakyReLU(0.01) -> Linear -> L2 normalise.

`_header` is an nn.Sequential with the reference's layout, so parameter names
(`_header.2.weight`, `_header.4.bias`, ...) and checkpoints interchange; its children only
hold parameters -- forward runs the HIP kernels (cyhip.functions.AvgPoolFn / LinearFn /
L2NormFn)."""
from __future__ import annotations

from torch import Tensor, nn

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from cyhip import ops
from cyhip.functions import (AdaptiveAvgPoolFn, AdaptiveMaxPoolFn, ClusterHeadFn, AvgPoolFn, DenseProjHiddenFn, GroupSoftmaxFn, HeadFn, L2NormFn,
                             LinearFn, ProjHeadFn)

from .nn import Flatten, Identical, Normalize, SoftmaxWithT

__all__ = ["ProjectionHead", "DenseProjectionHead", "ClusterHead", "DenseClusterHead"]


def _pair(v) -> Tuple[int, int]:
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class ProjectionHead(nn.Module):

    def __init__(self, *, input_dim: int, hidden_dim=256, output_dim: int, head_type: str, normalize: bool,
                 pool_name="adaptive_avg", spatial_size=(1, 1)):
        super().__init__()
        assert head_type in ("mlp", "linear"), head_type
        assert pool_name in ("adaptive_avg", "adaptive_max"), pool_name
        if tuple(_pair(spatial_size)) != (1, 1):
            # (the reference flattens the pooled map into nn.Linear(input_dim, ...): only (1, 1) has that many features)
            raise NotImplementedError("ProjectionHead pools to (1, 1): nn.Linear(input_dim, ...) follows the Flatten")
        self._input_dim, self._output_dim = input_dim, output_dim
        self._head_type, self._normalize = head_type, normalize
        self._pool_name = pool_name
        pool = nn.AdaptiveAvgPool2d((1, 1)) if pool_name == "adaptive_avg" else nn.AdaptiveMaxPool2d((1, 1))
        tail = Normalize() if normalize else Identical()
        if head_type == "mlp":
            self._header = nn.Sequential(pool, Flatten(), nn.Linear(input_dim, hidden_dim),
                                         nn.LeakyReLU(0.01, inplace=True), nn.Linear(hidden_dim, output_dim), tail)
        else:
            self._header = nn.Sequential(pool, Flatten(), nn.Linear(input_dim, output_dim), tail)

    def forward(self, features: Tensor) -> Tensor:
        h = self._header
        if (self._pool_name == "adaptive_avg" and self._head_type == "mlp" and self._normalize and h[2].bias is not None
                and h[4].bias is not None and ops.proj_head_ok(features, h[2].weight, h[4].weight)):
            return ProjHeadFn.apply(features, h[2].weight, h[2].bias, h[4].weight, h[4].bias)  # one launch
        x = AvgPoolFn.apply(features) if self._pool_name == "adaptive_avg" else AdaptiveMaxPoolFn.apply(features, (1, 1))
        if self._head_type == "mlp":
            x = LinearFn.apply(x, h[2].weight, h[2].bias, 1, 0.01)
            x = LinearFn.apply(x, h[4].weight, h[4].bias, 0, 0.0)
        else:
            x = LinearFn.apply(x, h[2].weight, h[2].bias, 0, 0.0)
        return L2NormFn.apply(x) if self._normalize else x


class DenseProjectionHead(nn.Module):
    """Pixel-wise projection head of the dense InfoNCE hook (contrastyou/projectors/heads.py:31-41,
    99-123): Conv1x1(C,hidden) -> LeakyReLU(0.01) -> Conv1x1(hidden,out) -> AdaptiveAvgPool2d(s) ->
    L2 normalise over channels.  `_projector` keeps the reference's Sequential layout (parameter
    names `_projector.0.weight`, `_projector.2.bias`, ...).

    Execution: the second 1x1 conv and the average pool are linear and commute, so the HIP path
    pools LeakyReLU(Conv1x1(x)) per output bin straight from the NHWC feature map (the
    [pixels, hidden] intermediate never exists) and applies the second conv to the pooled rows.
    `forward(features)` returns the full [N, out, s, s] map as the reference does;
    `project_points(features, points)` evaluates only the listed (image, row, col) bins -- what the
    dense hook needs after `region_extractor` -- and returns [len(points), out] rows."""

    def __init__(self, *, input_dim: int, hidden_dim=128, output_dim: int, head_type: str, normalize: bool,
                 pool_name="adaptive_avg", spatial_size=(16, 16)):
        super().__init__()
        assert head_type in ("mlp", "linear"), head_type
        assert pool_name in ("adaptive_avg", "adaptive_max", "identical", "none"), pool_name
        if pool_name in ("identical", "none", None):
            raise NotImplementedError("the HIP dense projector pools (adaptive_avg: the hook's setting, or adaptive_max)")
        self._input_dim, self._output_dim = input_dim, output_dim
        self._head_type, self._normalize = head_type, normalize
        self._pool_name, self._spatial_size = pool_name, _pair(spatial_size)
        self._pooling_module = (nn.AdaptiveAvgPool2d if pool_name == "adaptive_avg" else nn.AdaptiveMaxPool2d)(
            self._spatial_size)
        if head_type == "mlp":
            self._projector = nn.Sequential(nn.Conv2d(input_dim, hidden_dim, 1, 1, 0),
                                            nn.LeakyReLU(0.01, inplace=True),
                                            nn.Conv2d(hidden_dim, output_dim, 1, 1, 0))
        else:
            self._projector = nn.Sequential(nn.Conv2d(input_dim, output_dim, 1, 1, 0))

    def _rows_max(self, features: Tensor) -> Tensor:
        """pool_name="adaptive_max": a maximum does not commute with the second 1x1 conv, so the projector runs on every
        pixel (rows = pixels through the Linear kernels) and the [N, H, W, out] map is max-pooled per bin"""
        pr = self._projector
        x = ops.to_nhwc(features)
        N, Cc, H, W = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(N * H * W, Cc)
        if self._head_type == "mlp":
            rows = LinearFn.apply(rows, pr[0].weight.reshape(pr[0].weight.shape[0], -1), pr[0].bias, 1, 0.01)
            rows = LinearFn.apply(rows, pr[2].weight.reshape(pr[2].weight.shape[0], -1), pr[2].bias, 0, 0.0)
        else:
            rows = LinearFn.apply(rows, pr[0].weight.reshape(pr[0].weight.shape[0], -1), pr[0].bias, 0, 0.0)
        fmap = rows.view(N, H, W, -1).permute(0, 3, 1, 2)  # NHWC memory
        pooled = AdaptiveMaxPoolFn.apply(fmap, self._spatial_size)
        return L2NormFn.apply(pooled) if self._normalize else pooled

    def _rows(self, features: Tensor, bins: Optional[Tensor]) -> Tensor:
        pr = self._projector
        if self._pool_name == "adaptive_max":
            if bins is not None:
                raise NotImplementedError("point evaluation is implemented for adaptive_avg pooling")
            return self._rows_max(features)
        if self._head_type == "mlp":
            hp = DenseProjHiddenFn.apply(features, pr[0].weight, pr[0].bias, self._spatial_size, bins)
            w2 = pr[2].weight.reshape(pr[2].weight.shape[0], -1)
            rows = LinearFn.apply(hp, w2, pr[2].bias, 0, 0.0)
        else:
            if bins is not None:
                raise NotImplementedError("point evaluation is implemented for the mlp head")
            pooled = AdaptiveAvgPoolFn.apply(features, self._spatial_size)
            rows = LinearFn.apply(pooled, pr[0].weight.reshape(pr[0].weight.shape[0], -1), pr[0].bias, 0, 0.0)
        return L2NormFn.apply(rows) if self._normalize else rows

    def forward(self, features: Tensor) -> Tensor:
        ops.require_gpu(features)
        n = features.shape[0]
        sh, sw = self._spatial_size
        rows = self._rows(features, None)  # [n*sh*sw, out] == NHWC
        return rows.view(n, sh, sw, -1).permute(0, 3, 1, 2)

    def project_points(self, features: Tensor, points: Sequence[Sequence[Tuple[int, int]]]) -> Tensor:
        """points[i] = [(row, col), ...] on the s x s output grid of image i (distinct per image)
        -> [sum_i len(points[i]), out], image-major: == region_extractor(self(features))"""
        ops.require_gpu(features)
        bins = np.asarray([(i, a, b) for i, pts in enumerate(points) for a, b in pts], dtype=np.int32)
        assert len(np.unique(bins, axis=0)) == len(bins), "sampled bins must be distinct"
        return self._rows(features, ops._bins_tensor(bins, features.device))


def _sub_header(dense: bool, head_type: str, input_dim: int, hidden_dim: int, num_clusters: int, normalize: bool,
                T: float) -> nn.Sequential:
    tail = [Normalize() if normalize else Identical(), SoftmaxWithT(1, T=T)]
    if dense:
        if head_type == "linear":
            return nn.Sequential(nn.Conv2d(input_dim, num_clusters, 1, 1, 0), *tail)
        return nn.Sequential(nn.Conv2d(input_dim, hidden_dim, 1, 1, 0), nn.LeakyReLU(0.01, inplace=True),
                             nn.Conv2d(hidden_dim, num_clusters, 1, 1, 0), *tail)
    if head_type == "linear":
        return nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), Flatten(), nn.Linear(input_dim, num_clusters), *tail)
    return nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), Flatten(), nn.Linear(input_dim, 128),
                         nn.LeakyReLU(0.01, inplace=True), nn.Linear(128, num_clusters), *tail)


class _ClusterBase(nn.Module):
    _dense = False

    def __init__(self, *, input_dim: int, num_clusters: int, num_subheads: int, head_type: str, T: float,
                 normalize: bool, hidden_dim: int = 64):
        super().__init__()
        assert head_type in ("mlp", "linear"), head_type
        self._input_dim, self._output_dim = input_dim, num_clusters
        self._head_type, self._normalize = head_type, normalize
        self._num_clusters, self._num_subheads, self._T = num_clusters, num_subheads, T
        self._headers = nn.ModuleList([_sub_header(self._dense, head_type, input_dim, hidden_dim, num_clusters,
                                                   normalize, T) for _ in range(num_subheads)])

    def _normalise_groups(self, logits: Tensor) -> Tensor:
        """F.normalize(., dim=1) inside every sub-head's k outputs of stacked [M, S*k] logits"""
        M = logits.shape[0]
        return L2NormFn.apply(logits.reshape(M * self._num_subheads, self._num_clusters)).view(M, -1)

    def _logits(self, rows: Tensor, lin1: int, lin2: int) -> Tensor:
        """stacked [M, S*k] logits of all sub-heads on f32 rows [M, C] (pooled features, or pixels).
        linear: one stacked Linear; mlp: Linear(C, hidden) -> LeakyReLU(0.01) (stacked over the sub-heads, one
        launch) -> each sub-head's own Linear(hidden, k); then the optional per-sub-head L2 normalisation"""
        if self._head_type == "linear":
            w, b = self._stacked(lin1)
            out = LinearFn.apply(rows, w, b, 0, 0.0)
        else:
            w1 = torch.cat([h[lin1].weight.reshape(h[lin1].weight.shape[0], -1) for h in self._headers], dim=0)
            b1 = torch.cat([h[lin1].bias for h in self._headers], dim=0)
            hid = LinearFn.apply(rows, w1, b1, 1, 0.01)
            H = hid.shape[1] // self._num_subheads
            out = torch.cat([LinearFn.apply(hid[:, i * H:(i + 1) * H], h[lin2].weight.reshape(self._num_clusters, -1),
                                            h[lin2].bias, 0, 0.0) for i, h in enumerate(self._headers)], dim=1)
        return self._normalise_groups(out) if self._normalize else out

    def _stacked(self, idx: int):
        """all sub-heads as ONE [S*k, C] weight so the 1x1 conv / linear runs once"""
        w = torch.cat([h[idx].weight.reshape(self._num_clusters, -1) for h in self._headers], dim=0)
        b = torch.cat([h[idx].bias for h in self._headers], dim=0)
        return w, b


class ClusterHead(_ClusterBase):
    """IIC clustering head on encoder features (contrastyou/projectors/heads.py:44-62,125-148):
    per sub-head AdaptiveAvgPool2d(1) -> Linear(C,k) -> softmax(./T); returns a list of [n,k]."""

    def __init__(self, *, input_dim: int, num_clusters=5, num_subheads=10, head_type="linear", T=1, normalize=False):
        super().__init__(input_dim=input_dim, num_clusters=num_clusters, num_subheads=num_subheads,
                         head_type=head_type, T=T, normalize=normalize)

    def forward(self, features: Tensor) -> List[Tensor]:
        ops.require_gpu(features)
        pooled = AvgPoolFn.apply(features)
        logits = self._logits(pooled, lin1=2, lin2=4)
        probs = GroupSoftmaxFn.apply(logits, self._num_subheads, self._num_clusters, float(self._T))
        return list(probs.unbind(0))


class DenseClusterHead(_ClusterBase):
    """IIC segmentation clustering head on decoder features (projectors/heads.py:65-78,151-173):
    per sub-head Conv1x1(C,k) -> softmax over channels; returns a list of [n,k,H,W] (NHWC memory)."""
    _dense = True

    def __init__(self, *, input_dim: int, num_clusters=10, hidden_dim=64, num_subheads=10, T=1,
                 head_type: str = "linear", normalize: bool = False):
        super().__init__(input_dim=input_dim, num_clusters=num_clusters, num_subheads=num_subheads,
                         head_type=head_type, T=T, normalize=normalize, hidden_dim=hidden_dim)

    def forward(self, features: Tensor) -> List[Tensor]:
        ops.require_gpu(features)
        n, _, h, w_ = features.shape
        if (self._head_type == "linear" and not self._normalize
                and ops.cluster_head_ok(features.shape[1], self._num_subheads, self._num_clusters)):
            # conv1x1 + per-sub-head softmax in one pass on the matrix cores; the logits are never written
            w, b = self._stacked(0)
            rows = ops.to_nhwc(features).permute(0, 2, 3, 1).reshape(n * h * w_, -1)
            probs = ClusterHeadFn.apply(rows, w, b, self._num_subheads, self._num_clusters, float(self._T))
            return [p.view(n, h, w_, self._num_clusters).permute(0, 3, 1, 2) for p in probs.unbind(0)]
        if self._head_type == "linear":
            w, b = self._stacked(0)
            logits = HeadFn.apply(features, w.view(w.shape[0], w.shape[1], 1, 1), b)  # [n, S*k, h, w] NHWC f32
            flat = logits.permute(0, 2, 3, 1).reshape(n * h * w_, -1)
            if self._normalize:
                flat = self._normalise_groups(flat)
        else:  # per sub-head Conv1x1(C, hidden) -> LeakyReLU -> Conv1x1(hidden, k): pixels as rows of two linears
            rows = ops.to_nhwc(features).permute(0, 2, 3, 1).reshape(n * h * w_, -1).float()
            flat = self._logits(rows, lin1=0, lin2=2)
        probs = GroupSoftmaxFn.apply(flat, self._num_subheads, self._num_clusters, float(self._T))
        return [p.view(n, h, w_, self._num_clusters).permute(0, 3, 1, 2) for p in probs.unbind(0)]
