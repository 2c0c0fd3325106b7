"""HIP-graph replay of the network passes of a training step.

At the configured batch sizes the step is bound by the HOST: ~400 kernel launches, each behind a
few microseconds of Python (autograd Function, ctypes call, allocator), add up to ~9.5 ms per step
while the GPU needs less.  The two network passes of the two-stage step (labeled batch; unlabeled
batch + its transformed view, semi_seg/epochers/epocher.py:351-357) -- forward AND backward, with
the second-stream forks for the second pass and for the weight gradients inside -- are therefore
captured once into HIP graphs (torch.cuda.make_graphed_callables: one forward graph, one backward
graph, static input/output buffers, a private memory pool) and replayed every step.

What stays outside the graphs is everything that is data-dependent on the host: augmentation
parameters, label lists, the hooks' projectors and losses, the optimizer.  Forward hooks on the
model's blocks (the feature taps of the InfoNCE / MI hooks) do not fire inside a replay, so the
tapped block outputs are returned by the graphed callable -- with autograd history -- and the
registered hooks are called with them afterwards, in the order of the eager passes.

Which outputs receive a gradient depends on the registered hooks (in config C2 only the labeled
logits and the unlabeled Conv5 features do; the unlabeled decoder has no backward at all).  The first
step with a new configuration therefore runs eagerly and records, with tensor hooks, which outputs
the backward pass reached; the graphs are then captured with the other outputs detached, so the
captured backward contains exactly the eager one.  Detached outputs are returned behind a canary
autograd node that raises if a later loss does try to differentiate through them.

Conditions (checked per call; otherwise the eager two-stream path runs): every model parameter has
a live f32 `.grad` buffer that the kernels accumulate into (FusedRAdam's flat buffers), so that no
parameter gradient flows through autograd; bench instrumentation is off.  `CY_GRAPH_STEP=0`
disables the capture; a failed capture warns once and training continues eagerly.
"""
from __future__ import annotations

import os
import warnings
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
from torch import Tensor, nn

from . import functions as F
from . import ops

# On by default (CY_GRAPH_STEP=0 turns it off): the eager three-stream step is host-bound at 9.2-9.5
# ms/step and varies with the host; the replayed step is bound by the GPU's critical path (8.5 ms
# with this round's kernels) and every kernel-side gain shows up on it directly.
GRAPH_STEP = os.environ.get("CY_GRAPH_STEP", "1") != "0"


class _TwoPass(nn.Module):
    """model(xa) on the current stream, model(xb) on the "pass2" stream; returns both outputs and the
    outputs of the tapped blocks of both passes"""

    def __init__(self, model: nn.Module, taps: List[str], bn_context, need_grad=None):
        super().__init__()
        self.model = model
        self._taps = taps
        self._bn_context = bn_context
        self._need_grad = need_grad  # per output: keep its autograd history? (None: all)

    def forward(self, xa: Tensor, xb: Tensor):
        dev = xa.device
        main = torch.cuda.current_stream(dev)
        side = ops.side_stream(dev, "pass2")
        grabbed: List[Tensor] = []
        handles = [self.model.get_module(n).register_forward_hook(lambda m, i, o: grabbed.append(o))
                   for n in self._taps]
        try:
            if ops.TWO_STREAM:
                side.wait_stream(main)
                ops.note_side_work(side)
                ya = self.model(xa)
                with torch.cuda.stream(side), self._bn_context(self.model):
                    yb = self.model(xb)
                main.wait_stream(side)
            else:
                ya = self.model(xa)
                with self._bn_context(self.model):
                    yb = self.model(xb)
        finally:
            for h in handles:
                h.remove()
        outs = (ya, yb, *grabbed)
        if self._need_grad is not None:
            outs = tuple(o if k else o.detach() for o, k in zip(outs, self._need_grad))
        return outs


class _Canary(torch.autograd.Function):
    """identity on an output the graphs were captured WITHOUT a backward for; differentiating through
    it means the loss configuration changed after the capture"""

    @staticmethod
    def forward(ctx, x: Tensor, anchor: Tensor, owner):
        ctx.owner = owner
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.owner.invalid = True
        raise RuntimeError("cyhip.graphed: a loss now differentiates through a network output that received no "
                           "gradient when the step was captured; the graphs are dropped and rebuilt on the next "
                           "step (set CY_GRAPH_STEP=0 to train eagerly)")


def _tapped_blocks(model: nn.Module) -> List[str]:
    return [n for n in model.arch_elements if len(model.get_module(n)._forward_hooks) > 0]


def _sinks_ready(model: nn.Module) -> bool:
    for p in model.parameters():
        if not p.requires_grad:
            continue
        g = p.grad
        if g is None or not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous():
            return False
    return True


class GraphedTwoPass:
    """build once per (model storage, shapes, taps, BN flags), then `ya, yb = graphed(xa, xb)`"""

    def __init__(self, model: nn.Module, bn_context, xa: Tensor, xb: Tensor, autocast_dtype, need_grad):
        self.model = model
        self.taps = _tapped_blocks(model)
        self.need_grad = tuple(need_grad)
        self.invalid = False
        self._anchor = torch.zeros((), device=xa.device, requires_grad=True)
        self._wrapper = _TwoPass(model, self.taps, bn_context, self.need_grad)
        self._wrapper.train(model.training)
        params = [p for p in model.parameters() if p.requires_grad]
        state = {k: v.clone() for k, v in model.state_dict().items()}
        # warm-up and capture run the passes for real and accumulate garbage into the live .grad buffers.
        # Inside a gradient-accumulation window (AMPScaler.optimizer_zero only zeroes every
        # accumulate_iter steps, contrastyou/amp/amp.py:33-39) those buffers hold the previous
        # micro-batches' gradients: snapshot them (and the "received a gradient" flags) and put them back.
        grads_before = [p.grad.detach().clone() for p in params]
        touched_before = [bool(p.__dict__.get("_cy_touched", False)) for p in params]
        silence = _silenced_taps(model, self.taps)  # the real taps must not see warm-up / capture tensors
        silence.__enter__()
        for p in params:
            p.__dict__["_cy_touched"] = False
        ops.CAPTURING = True
        # inside a graph the weight-gradient kernels stay on their pass's branch: the runtime overlaps
        # the two long pass branches well, but per-layer forks to a third branch cost more in
        # dependency edges than they overlap (8.5 vs 9.1 ms/step, same box)
        async_wgrad, ops.ASYNC_WGRAD = ops.ASYNC_WGRAD, False
        F.bump_weights_epoch()  # the weight packing kernels must be part of the captured forward
        try:
            ctx = (torch.autocast("cuda", dtype=autocast_dtype, cache_enabled=False) if autocast_dtype is not None
                   else torch.autocast("cuda", enabled=False))
            with ctx:
                self._graphed = torch.cuda.make_graphed_callables(self._wrapper, (xa, xb), allow_unused_input=True)
            self._touched = [p for p in params if p.__dict__.get("_cy_touched")]
            # the capture recorded the "gradients final" marks as external event nodes: every replay records them
            self._marks = ops.take_ready_marks()
        finally:
            ops.CAPTURING = False
            ops.ASYNC_WGRAD = async_wgrad
            F.bump_weights_epoch()
            silence.__exit__()
            # warm-up and capture ran the passes for real (also when the capture failed half-way): put
            # running statistics / counters and the gradient buffers back
            torch.cuda.synchronize()
            with torch.no_grad():
                model.load_state_dict(state, strict=True)
                torch._foreach_copy_([p.grad for p in params], grads_before)
            for p, t in zip(params, touched_before):
                p.__dict__["_cy_touched"] = t

    def __call__(self, xa: Tensor, xb: Tensor) -> Tuple[Tensor, Tensor]:
        ops.note_home_stream(xa.device)
        ops.begin_step_marks(xa.device)  # (data parallel: the step's serial, read by the mark copies of the replay)
        out = self._graphed(xa, xb)
        out = tuple(o if k else _Canary.apply(o, self._anchor, self) for o, k in zip(out, self.need_grad))
        ya, yb, feats = out[0], out[1], out[2:]
        for p in self._touched:  # what ops.grad_sink does on the eager path
            p.__dict__["_cy_touched"] = True
        marks = getattr(self, "_marks", None)
        if marks:
            # the backward graph leaves the marks of its capture again -- once it replays: they count from the moment
            # a gradient arrives at the passes' outputs (a consumer waiting for a mark that never comes would hang)
            for o in (ya, yb):
                if o.requires_grad:
                    o.register_hook(lambda g, m=marks: (ops.set_ready_marks(m), g)[1])
        _fire_taps(self.model, self.taps, feats)
        return ya, yb


def _fire_taps(model: nn.Module, taps: List[str], feats) -> None:
    """hand the tapped blocks' outputs to the registered forward hooks in the eager order: the blocks
    of the labeled pass, then those of the unlabeled pass"""
    k = len(taps)
    for pass_idx in range(2):
        for j, n in enumerate(taps):
            m = model.get_module(n)
            for hook in list(m._forward_hooks.values()):
                hook(m, (None,), feats[pass_idx * k + j])


class _silenced_taps:
    """`with _silenced_taps(model, taps):` -- the registered forward hooks of the tapped blocks see nothing"""

    def __init__(self, model: nn.Module, taps: List[str]):
        self.blocks = [model.get_module(n) for n in taps]

    def __enter__(self):
        self.stash = [m._forward_hooks for m in self.blocks]
        for m in self.blocks:
            m._forward_hooks = OrderedDict()

    def __exit__(self, *exc):
        for m, hooks in zip(self.blocks, self.stash):
            m._forward_hooks = hooks
        return False


def _key(model: nn.Module, xa: Tensor, xb: Tensor, disable_bn: bool, autocast_dtype) -> tuple:
    p0 = next(model.parameters())
    flags = tuple(bool(m.track_running_stats) for m in model.modules() if isinstance(m, nn.BatchNorm2d))
    req = tuple(p.requires_grad for p in model.parameters())
    return (tuple(xa.shape), tuple(xb.shape), xa.dtype, xb.dtype, p0.data_ptr(),
            p0.grad.data_ptr() if p0.grad is not None else 0, tuple(_tapped_blocks(model)), flags, req,
            ops.TWO_STREAM,
            bool(disable_bn), model.training, autocast_dtype, getattr(model, "compute_dtype", None))


_failed = False


class _Probe:
    """one eager step that records which outputs of the two passes the backward pass reaches"""

    def __init__(self, n_out: int):
        self.reached = [False] * n_out
        self.ran_backward = False

    def watch(self, outs):
        for i, o in enumerate(outs):
            if o.requires_grad:
                o.register_hook(lambda g, i=i: self._hit(i))

    def _hit(self, i):
        self.reached[i] = True
        self.ran_backward = True


def two_pass(model: nn.Module, bn_context, xa: Tensor, xb: Tensor, disable_bn: bool, autocast_dtype):
    """the two passes of the step: graph replay when a capture for this configuration exists, an eager
    two-stream probe step before that; None when capture is not applicable (the caller runs eagerly)"""
    global _failed
    if not GRAPH_STEP or _failed or ops.PROFILE is not None:
        return None
    if not _sinks_ready(model) or not torch.is_grad_enabled():
        return None
    cache: Dict[tuple, object] = model.__dict__.setdefault("_cy_graphed", {})
    key = _key(model, xa, xb, disable_bn, autocast_dtype)
    entry = cache.get(key)
    if isinstance(entry, GraphedTwoPass):
        if not entry.invalid:
            return entry(xa, xb)
        entry = None
    if isinstance(entry, _Probe) and entry.ran_backward:
        try:
            g = GraphedTwoPass(model, bn_context, xa, xb, autocast_dtype, entry.reached)
        except Exception as e:  # capture is an optimisation: say so once and keep training eagerly
            _failed = True
            warnings.warn(f"HIP-graph capture of the training passes failed ({type(e).__name__}: {e}); "
                          "continuing with eager launches")
            return None
        cache[key] = g
        return g(xa, xb)
    # probe step: the eager two-stream passes, with gradient-flow detection on every output
    taps = _tapped_blocks(model)
    ops.note_home_stream(xa.device)
    with _silenced_taps(model, taps):
        outs = _TwoPass(model, taps, bn_context)(xa, xb)
    # the hooks get aliases of the tapped outputs: a gradient arriving at an alias came from a hook,
    # not from the network's own path through that block (which every tapped block of a pass that is
    # differentiated at all would otherwise report)
    outs = outs[:2] + tuple(o.view_as(o) for o in outs[2:])
    probe = _Probe(len(outs))
    probe.watch(outs)
    _fire_taps(model, taps, outs[2:])
    if len(cache) > 4:
        cache.clear()
    cache[key] = probe
    return outs[0], outs[1]
