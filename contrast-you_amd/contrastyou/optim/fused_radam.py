"""Flat-buffer RAdam.

MI355X-first layout: all parameters of a group live in one contiguous f32 buffer (each
nn.Parameter is a view into it) and all their gradients in another, so that
  * the optimizer step is one streaming kernel (cy_radam_step) instead of ~70 small ones,
  * the data-parallel gradient reduction is ONE RCCL all-reduce over xGMI per group
    (35 MB for the U-Net) instead of one per tensor,
  * an EMA teacher update is one kernel too (semi_seg.hooks.mt).
Semantics follow torch.optim.RAdam (decoupled_weight_decay=False): see csrc/cy_misc.hip.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from cyhip import ops
from cyhip.functions import bump_weights_epoch


class FlatParams:
    """Re-point `params` (and their .grad) into flat f32 buffers."""

    def __init__(self, params: List[nn.Parameter]):
        params = [p for p in params]
        assert params, "empty parameter list"
        dev = params[0].device
        total = sum(p.numel() for p in params)
        self.params = params
        self.data = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            assert p.dtype == torch.float32 and p.device == dev, "flat buffers hold f32 parameters of one device"
            self.data[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            off += n
        self.numel = total

    def zero_grad(self):
        self.grad.zero_()
        off = 0
        for p in self.params:  # re-attach views someone may have dropped (set_to_none)
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + n].view(p.shape)
            off += n


class FusedRAdam(torch.optim.Optimizer):

    def __init__(self, params: Iterable, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 process_group: Optional["dist.ProcessGroup"] = None, data_parallel: Optional[bool] = None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._pg = process_group
        if data_parallel is None:
            data_parallel = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self._dp = data_parallel
        self._flat: List[Optional[FlatParams]] = [None] * len(self.param_groups)
        self._flat_state = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_flat"):
            self._flat.append(None)

    def _ensure_flat(self):
        for i, g in enumerate(self.param_groups):
            if self._flat[i] is None:
                fp = FlatParams(list(g["params"]))
                self._flat[i] = fp
                self._flat_state[i] = dict(step=0, exp_avg=torch.zeros_like(fp.data),
                                           exp_avg_sq=torch.zeros_like(fp.data))
        bump_weights_epoch()

    def zero_grad(self, set_to_none: bool = False):
        if any(f is None for f in self._flat):
            self._ensure_flat()
        for f in self._flat:
            f.zero_grad()

    def all_reduce_grads(self):
        """mean over data-parallel ranks; one collective per parameter group"""
        if not self._dp:
            return
        world = dist.get_world_size(self._pg)
        for f in self._flat:
            if f.grad.is_cuda:
                dist.all_reduce(f.grad, op=dist.ReduceOp.AVG, group=self._pg)
            else:
                dist.all_reduce(f.grad, op=dist.ReduceOp.SUM, group=self._pg)
                f.grad.div_(world)

    def state_dict(self):
        """{"param_groups": [...], "flat": {group: {step, exp_avg, exp_avg_sq}}}"""
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return {"param_groups": groups, "flat": {i: dict(st) for i, st in self._flat_state.items()}}

    def load_state_dict(self, state):
        self._ensure_flat()
        for g, sg in zip(self.param_groups, state["param_groups"]):
            g.update(sg)
        for i, st in state["flat"].items():
            mine = self._flat_state[int(i)]
            mine["step"] = int(st["step"])
            mine["exp_avg"].copy_(st["exp_avg"])
            mine["exp_avg_sq"].copy_(st["exp_avg_sq"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if any(f is None for f in self._flat):
            self._ensure_flat()
        self.all_reduce_grads()
        for i, g in enumerate(self.param_groups):
            f, st = self._flat[i], self._flat_state[i]
            st["step"] += 1
            ops.radam_step(f.data, f.grad, st["exp_avg"], st["exp_avg_sq"], g["lr"], g["betas"][0], g["betas"][1],
                           g["eps"], g["weight_decay"], st["step"])
        bump_weights_epoch()
        return loss
