"""`UniversalDice` (contrastyou/meters/general_dice_meter.py:17-127): per-group (scan) accumulated
intersection / union of one-hot prediction and target; dice = (2I + 1e-16)/(U + 1e-16), mean
over groups, `DSC_mean` over the reported classes.

On the GPU path `add_logits(logits, target, group_name)` takes the arg-max inside the HIP
`cy_dice_counts` kernel and leaves the integer counts on the device; they are only read back
in `summary()`, so the training loop has no per-batch sync.  `add(pred, target, ...)` keeps
the reference signature (class- or one-hot-coded tensors, any device).
"""
from __future__ import annotations

import typing as t
from collections import OrderedDict

import numpy as np
import torch
from torch import Tensor

from ..types import to_float
from ..utils.general import class2one_hot, one_hot, probs2one_hot, simplex
from ..utils.utils import average_iter
from .metric import Metric


class UniversalDice(Metric):
    def __init__(self, C: int, report_axis: t.Iterable[int] = None) -> None:
        super().__init__()
        if report_axis is not None:
            report_axis = list(report_axis)
            assert max(report_axis) <= C, f"Incompatible parameter of `C`={C} and `report_axises`={report_axis}"
        self._C = C
        self._report_axis = list(range(C)) if report_axis is None else report_axis
        self.reset()

    def reset(self):
        self._intersections: "OrderedDict[str, Tensor]" = OrderedDict()
        self._unions: "OrderedDict[str, Tensor]" = OrderedDict()
        self._pending: t.List[t.Tuple[Tensor, t.List[str]]] = []
        self._n = 0

    def _names(self, group_name, B: int) -> t.List[str]:
        if group_name is None:
            return [f"{self._n}_{i:03d}" for i in range(B)]
        if isinstance(group_name, str):
            return [group_name] * B
        if isinstance(group_name, (tuple, list)):
            assert len(group_name) == B
            return list(group_name)
        raise TypeError(f"type of `group_name` wrong {type(group_name)}")

    def _accumulate(self, inter: Tensor, union: Tensor, names: t.List[str]):
        for i_, u_, g in zip(inter, union, names):
            if g in self._intersections:
                self._intersections[g] = self._intersections[g] + i_
                self._unions[g] = self._unions[g] + u_
            else:
                self._intersections[g], self._unions[g] = i_, u_

    @torch.no_grad()
    def add_logits(self, logits: Tensor, target: Tensor, *, group_name=None):
        """fused arg-max + counting on the device (HIP); no host sync"""
        from cyhip import ops
        tgt = target.squeeze(1) if target.dim() == 4 else target
        counts = ops.dice_counts(ops.to_nhwc(logits.detach().float()), tgt.contiguous())
        self._pending.append((counts, self._names(group_name, logits.shape[0])))
        self._n += 1

    @torch.no_grad()
    def _add(self, pred: Tensor, target: Tensor, *, group_name=None):
        assert pred.shape == target.shape, f"incompatible shape of `pred` and `target`, given {pred.shape} and {target.shape}."
        pred, target = pred.detach(), target.detach()
        if pred.dim() >= 3 and pred.is_floating_point() and simplex(pred, 1) and one_hot(target):
            po, to = probs2one_hot(pred).long(), target.long()
        else:
            po, to = class2one_hot(pred, self._C).long(), class2one_hot(target, self._C).long()
        dims = list(range(2, po.dim()))
        self._accumulate((po * to).sum(dims).cpu(), (po + to).sum(dims).cpu(), self._names(group_name, pred.shape[0]))
        self._n += 1

    def _flush(self):
        for counts, names in self._pending:
            c = counts.cpu()
            self._accumulate(c[..., 0], c[..., 1], names)
        self._pending = []

    def compute_dice_by_group(self) -> t.Optional[Tensor]:
        self._flush()
        if self._n > 0:
            inter = torch.stack(tuple(self._intersections.values()), dim=0)
            union = torch.stack(tuple(self._unions.values()), dim=0)
            return (2 * inter.float() + 1e-16) / (union.float() + 1e-16)

    def _summary(self):
        if self._n > 0:
            means = self.compute_dice_by_group().mean(dim=0)
        else:
            means = (np.nan,) * self._C
        report = {f"DSC{i}": to_float(means[i]) for i in self._report_axis}
        report["DSC_mean"] = average_iter(report.values())
        return report

    @property
    def group_names(self):
        self._flush()
        return sorted(self._intersections.keys())
