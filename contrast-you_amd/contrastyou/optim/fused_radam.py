"""Flat-buffer RAdam.

MI355X-first layout: all parameters of a group live in one contiguous f32 buffer (each
nn.Parameter is a view into it) and all their gradients in another, so that
  * the optimizer step is one streaming kernel (cy_radam_step) instead of ~70 small ones,
  * the data-parallel gradient reduction is ONE RCCL all-reduce over xGMI per group
    (35 MB for the U-Net) instead of one per tensor,
  * an EMA teacher update is one kernel too (semi_seg.hooks.mt).
Semantics follow torch.optim.RAdam (decoupled_weight_decay=False): see csrc/cy_misc.hip.
Like torch's optimizer, a parameter that received no gradient since the last zero_grad() is
skipped (no weight decay, no moment update, its step count does not advance) -- e.g. the decoder
during encoder pre-training.  "Received a gradient" is tracked per parameter (autograd's
post-accumulate hook, or the kernels' direct accumulation through ops.grad_sink); the step is one
kernel launch per run of consecutive updated parameters with the same step count (normally one).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from cyhip import ops
from cyhip.functions import bump_weights_epoch


def _mark_touched(p):
    p.__dict__["_cy_touched"] = True


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
            p.register_post_accumulate_grad_hook(_mark_touched)
            off += n
        self.numel = total
        self.offsets = [0]
        for p in params:
            self.offsets.append(self.offsets[-1] + p.numel())

    def stale(self) -> bool:
        """a parameter was moved / re-pointed (e.g. module.to(device)) after flattening"""
        lo, hi = self.data.data_ptr(), self.data.data_ptr() + 4 * self.numel
        return any(p.device != self.data.device or not (lo <= p.data_ptr() < hi or p.numel() == 0)
                   for p in self.params)

    def touched(self) -> List[bool]:
        return [bool(p.__dict__.get("_cy_touched", False)) for p in self.params]

    def clear_touched(self):
        for p in self.params:
            p.__dict__["_cy_touched"] = False

    def zero_grad(self):
        self.grad.zero_()
        self.clear_touched()
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
        """(re)build the flat buffers lazily: at the first zero_grad()/step(), and again if the
        parameters were moved to another device afterwards (moments follow them)"""
        for i, g in enumerate(self.param_groups):
            old = self._flat[i]
            if old is not None and not old.stale():
                continue
            fp = FlatParams(list(g["params"]))
            self._flat[i] = fp
            st = self._flat_state.get(i)
            if st is None:
                st = dict(step=0, steps=[0] * len(fp.params), exp_avg=torch.zeros_like(fp.data),
                          exp_avg_sq=torch.zeros_like(fp.data))
            else:
                st["exp_avg"] = st["exp_avg"].to(fp.data.device)
                st["exp_avg_sq"] = st["exp_avg_sq"].to(fp.data.device)
            self._flat_state[i] = st
        pending, self._pending = getattr(self, "_pending", None), None
        if pending is not None:
            self._apply_state(pending)
        self.broadcast_state()
        bump_weights_epoch()

    def broadcast_state(self, src: int = 0):
        """data-parallel replicas must start from identical parameters and moments (the gradient all-reduce
        keeps them identical from then on): what DistributedDataParallel's constructor does for the
        reference's plain modules.  One broadcast per flat buffer; called whenever the flat buffers are
        (re)built and after a checkpoint has been applied."""
        if not self._dp:
            return
        src_global = dist.get_global_rank(self._pg, src) if self._pg is not None else src
        for i, f in enumerate(self._flat):
            if f is None:
                continue
            st = self._flat_state[i]
            for buf in (f.data, st["exp_avg"], st["exp_avg_sq"]):
                dist.broadcast(buf, src=src_global, group=self._pg)
            meta = torch.tensor([st["step"], *st["steps"]], dtype=torch.int64, device=f.data.device)
            dist.broadcast(meta, src=src_global, group=self._pg)
            vals = meta.tolist()
            st["step"], st["steps"] = int(vals[0]), [int(v) for v in vals[1:]]

    def _needs_flat(self) -> bool:
        return any(f is None or f.stale() for f in self._flat)

    def zero_grad(self, set_to_none: bool = False):
        ops.join_side_streams(torch.cuda.current_stream() if torch.cuda.is_available() else None)
        if self._needs_flat():
            self._ensure_flat()
        self._reduced = False
        for f in self._flat:
            f.zero_grad()

    def all_reduce_grads(self):
        """mean over data-parallel ranks; one collective per parameter group"""
        if not self._dp or getattr(self, "_reduced", False):
            return
        if self._needs_flat():
            self._ensure_flat()
        if torch.cuda.is_available():
            ops.join_side_streams(torch.cuda.current_stream())
        self._reduced = True
        world = dist.get_world_size(self._pg)
        for f in self._flat:
            if f.grad.is_cuda:
                dist.all_reduce(f.grad, op=dist.ReduceOp.AVG, group=self._pg)
            else:
                dist.all_reduce(f.grad, op=dist.ReduceOp.SUM, group=self._pg)
                f.grad.div_(world)

    def state_dict(self):
        """{"param_groups": [...], "flat": {group: {step, steps, exp_avg, exp_avg_sq}}}"""
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        pending = getattr(self, "_pending", None)
        if pending is not None:  # loaded but not stepped yet: hand the loaded moments back
            return {"param_groups": groups, "flat": pending["flat"]}
        return {"param_groups": groups, "flat": {i: dict(st) for i, st in self._flat_state.items()}}

    def load_state_dict(self, state):
        """hyper-parameters apply at once; the moments are copied when the flat buffers exist on the
        parameters' final device (a Trainer loads checkpoints before `.to(device)`)"""
        for g, sg in zip(self.param_groups, state["param_groups"]):
            g.update(sg)
        self._pending = state
        if not self._needs_flat():
            self._pending = None
            self._apply_state(state)
            self.broadcast_state()

    def _apply_state(self, state):
        for i, st in state["flat"].items():
            mine = self._flat_state[int(i)]
            mine["step"] = int(st["step"])
            mine["steps"] = [int(v) for v in st.get("steps", [int(st["step"])] * len(mine["steps"]))]
            mine["exp_avg"].copy_(st["exp_avg"])
            mine["exp_avg_sq"].copy_(st["exp_avg_sq"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._needs_flat():
            self._ensure_flat()
        if torch.cuda.is_available():  # weight gradients are produced on a side stream
            ops.join_side_streams(torch.cuda.current_stream())
        self.all_reduce_grads()
        self._reduced = False
        for i, g in enumerate(self.param_groups):
            f, st = self._flat[i], self._flat_state[i]
            st["step"] += 1
            steps, touched = st["steps"], f.touched()
            j, n = 0, len(steps)
            while j < n:  # runs of consecutive updated parameters that share a step count
                if not touched[j]:
                    j += 1
                    continue
                e = j
                while e + 1 < n and touched[e + 1] and steps[e + 1] == steps[j]:
                    e += 1
                a, b = f.offsets[j], f.offsets[e + 1]
                ops.radam_step(f.data[a:b], f.grad[a:b], st["exp_avg"][a:b], st["exp_avg_sq"][a:b], g["lr"],
                               g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], steps[j] + 1)
                for q in range(j, e + 1):
                    steps[q] += 1
                j = e + 1
        bump_weights_epoch()
        return loss
