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
        if data_parallel:
            ops.marks_wanted = True
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
        if self._dp and not ops.DP_EARLY_FORCED:
            # early start of the marked buckets: by rule, from the gradient volume and the world size (ops.dp_early_rule)
            nbytes = 4 * sum(f.numel for f in self._flat if f is not None)
            ops.DP_EARLY = ops.dp_early_rule(nbytes, dist.get_world_size(self._pg))
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

    def broadcast_buffers(self, *modules, src: int = 0):
        """rank `src`'s module buffers (BatchNorm running statistics, batch counters) to every replica -- the other
        half of what DistributedDataParallel's constructor broadcasts; called by the trainer after a resume / before
        the first epoch (COLLECTIVE, like `broadcast_state`)"""
        if not self._dp:
            return
        src_global = dist.get_global_rank(self._pg, src) if self._pg is not None else src
        for m in modules:
            for b in m.buffers():
                dist.broadcast(b, src=src_global, group=self._pg)

    def _needs_flat(self) -> bool:
        return any(f is None or f.stale() for f in self._flat)

    def zero_grad(self, set_to_none: bool = False):
        ops.join_side_streams(torch.cuda.current_stream() if torch.cuda.is_available() else None)
        if self._needs_flat():
            self._ensure_flat()
        self._reduced = False
        for f in self._flat:
            f.zero_grad()

    # One collective per parameter group at the U-Net's size (8.6 M parameters = 34.5 MB of f32): nothing overlaps the
    # collectives but the 40 us of RAdam launches, so fewer, larger messages win (one-rank RCCL, tools/dp_single_rank.py:
    # 6.72 ms/step with 8 MB buckets, 6.62 with one bucket -- launch overhead alone; larger messages also use the links
    # better).  Models beyond 64 MB of gradients are cut into several.
    BUCKET_ELEMS = 16 << 20

    def _buckets(self, f: "FlatParams"):
        """parameter-aligned slices [a, b) of a flat buffer, in REVERSE parameter order (the order the backward
        pass completes gradients in: 1x1 head and decoder first, Conv1 last), ~BUCKET_ELEMS each"""
        cuts = []
        j = len(f.params)
        # With the early start on (ops.DP_EARLY) a bucket never mixes parameters of different "gradients final" marks
        # (decoder / conv5 / conv4 / unmarked: contrastyou/arch/unet.py) -- one bucket over the whole group would hold the
        # unmarked Conv1-3, head and projector parameters too and could never start early.
        tag = (lambda q: f.params[q].__dict__.get("_cy_ready_tag")) if ops.DP_EARLY else (lambda q: None)
        while j > 0:
            i = j
            while (i > 0 and f.offsets[j] - f.offsets[i - 1] <= self.BUCKET_ELEMS
                   and tag(i - 1) == tag(j - 1)):
                i -= 1
            if i == j:  # a single parameter larger than a bucket
                i = j - 1
            cuts.append((f.offsets[i], f.offsets[j], i, j))
            j = i
        return cuts

    def all_reduce_grads(self, wait: bool = True):
        """gradient mean over the data-parallel ranks: bucketed, asynchronous collectives over xGMI (RCCL) in the
        order the backward pass finishes the gradients; `wait=False` leaves the handles for `step()`, which
        consumes bucket k (RAdam on its parameters) while bucket k+1 is still on the wire"""
        if not self._dp or getattr(self, "_reduced", False):
            return
        if self._needs_flat():
            self._ensure_flat()
        if torch.cuda.is_available():
            ops.join_side_streams(torch.cuda.current_stream())
        self._reduced = True
        world = dist.get_world_size(self._pg)
        avg = dist.get_backend(self._pg) == "nccl"  # (gloo has no AVG: sum, then scale)
        self._inflight = []
        # Buckets whose parameters all carry a "gradients final" mark of this step (the decoder, Conv5, Conv4:
        # contrastyou/arch/unet.py) are reduced on a communication stream that waits for those marks only -- the
        # rest of the backward pass (eager or a replayed graph) is still running then.  Everything else waits for
        # the whole pass, as before.
        marks = ops.take_ready_marks() if torch.cuda.is_available() else {}
        comm = None
        self.dp_steps = getattr(self, "dp_steps", 0) + 1
        for gi, f in enumerate(self._flat):
            for a, b, i, j in self._buckets(f):
                view = f.grad[a:b]
                tags = {p.__dict__.get("_cy_ready_tag") for p in f.params[i:j]}
                early = bool(marks) and None not in tags
                if early:
                    # a mark covers the blocks that finish before it (ops.MARK_TAGS is in backward order): wait for
                    # the first mark left at or after the latest block of the bucket
                    last = max(ops.MARK_TAGS.index(t) for t in tags)
                    cover = [t for t in ops.MARK_TAGS[last:] if t in marks]
                    early = bool(cover)
                    tags = {cover[0]} if cover else tags
                if early:
                    if comm is None:
                        comm = ops.side_stream(view.device, "comm")
                    ops.stream_wait_marks(comm, [marks[t] for t in tags], view.device)
                    with torch.cuda.stream(comm):
                        h = dist.all_reduce(view, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM,
                                            group=self._pg, async_op=True)
                    self.early_buckets = getattr(self, "early_buckets", 0) + 1
                else:
                    h = dist.all_reduce(view, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self._pg,
                                        async_op=True)
                self.dp_bytes = getattr(self, "dp_bytes", 0) + 4 * view.numel()
                self.dp_buckets = getattr(self, "dp_buckets", 0) + 1
                self._inflight.append((gi, i, j, view, h, None if avg else world))
        if wait:
            self._wait_inflight()

    def _wait_inflight(self, upto=None):
        """wait for (and finish) the collectives in flight; `upto` = (group, first param) stops after that bucket"""
        pend = getattr(self, "_inflight", None) or []
        while pend:
            gi, i, j, view, h, div = pend.pop(0)
            h.wait()
            if div:
                view.div_(div)
            if upto is not None and (gi, i) == upto:
                break
        self._inflight = pend

    def state_dict(self):
        """{"param_groups": [...], "flat": {group: {step, steps, exp_avg, exp_avg_sq}}}"""
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        pending = getattr(self, "_pending", None)
        if pending is not None:  # loaded but not stepped yet: hand the loaded moments back
            return {"param_groups": groups, "flat": pending["flat"]}
        return {"param_groups": groups, "flat": {i: dict(st) for i, st in self._flat_state.items()}}

    def load_state_dict(self, state):
        """hyper-parameters apply at once; the moments are copied when the flat buffers exist on the
        parameters' final device (a Trainer loads checkpoints before `.to(device)`).
        Data parallel: COLLECTIVE -- every rank must call it (rank 0's moments and parameters are then broadcast,
        so only rank 0's checkpoint content matters); a call on one rank alone would leave its broadcasts without
        partners.  Module buffers (BN running statistics) travel with `broadcast_buffers(*modules)`."""
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
        self.all_reduce_grads(wait=False)
        self._reduced = False
        # buckets in the order their collectives were launched (reverse parameter order); without data parallelism
        # one "bucket" per group.  Runs of consecutive updated parameters that share a step count -> one launch.
        plan = []
        for gi, f in enumerate(self._flat):
            plan += [(gi, i, j) for _, _, i, j in self._buckets(f)] if self._dp else [(gi, 0, len(f.params))]
        for gi in range(len(self.param_groups)):
            self._flat_state[gi]["step"] += 1
        touched_all = [f.touched() for f in self._flat]
        for gi, lo, hi in plan:
            if self._dp:
                self._wait_inflight(upto=(gi, lo))
            g, f, st = self.param_groups[gi], self._flat[gi], self._flat_state[gi]
            steps, touched = st["steps"], touched_all[gi]
            j = lo
            while j < hi:
                if not touched[j]:
                    j += 1
                    continue
                e = j
                while e + 1 < hi and touched[e + 1] and steps[e + 1] == steps[j]:
                    e += 1
                a, b = f.offsets[j], f.offsets[e + 1]
                ops.radam_step(f.data[a:b], f.grad[a:b], st["exp_avg"][a:b], st["exp_avg_sq"][a:b], g["lr"],
                               g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], steps[j] + 1)
                for q in range(j, e + 1):
                    steps[q] += 1
                j = e + 1
        self._wait_inflight()
        bump_weights_epoch()
        return loss
