"""torch.autograd.Function wrappers: the seam between the reference's Python object
protocol (nn.Module blocks, forward hooks, loss objects) and the HIP kernels.

Granularity follows the reference's named blocks (contrastyou/arch/unet.py:16-46): one
Function per _ConvBlock / _UpConv so that `get_module(name)` forward hooks still observe a
block-output tensor with autograd history, while inside a block nothing but raw conv
outputs is materialised (BN+ReLU of conv 1 is applied in conv 2's load path; max-pool,
nearest-upsample and channel concat are addressing modes of the consuming conv).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
from torch import Tensor, nn

from . import ops

# bumped by anything that rewrites parameter memory without going through torch
# (fused optimizer, flat-buffer EMA): invalidates the packed-weight cache.
_weights_epoch = 0


# num_batches_tracked counters of the BN layers touched by the current U-Net forward: bumped by
# ONE multi-tensor add at the end of the forward instead of one tiny kernel per BN layer.
_nbt_pending: Optional[list] = None


# running statistics of the BN layers touched by the current U-Net forward (accumulator path, csrc/cy_bn_acc.h): their
# consumers leave the batch moments in memory, ONE launch at the end of the forward blends them into the running buffers
_run_pending: Optional[list] = None


class defer_batch_counters:
    """scope of one network pass: the BatchNorm accumulators of the pass come from one zeroed arena (one fill launch),
    the running statistics and batch counters of its layers are updated by one launch each when the pass ends"""

    def __init__(self, device=None):
        self._device = device

    def __enter__(self):
        global _nbt_pending, _run_pending
        self._outer = (_nbt_pending, _run_pending)
        _nbt_pending, _run_pending = [], []
        self._arena = ops.bn_arena_begin(self._device) if self._device is not None else None
        return self

    def __exit__(self, *exc):
        global _nbt_pending, _run_pending
        pending, running = _nbt_pending, _run_pending
        _nbt_pending, _run_pending = self._outer
        if self._device is not None:
            ops.bn_arena_end(self._arena)
        if exc[0] is None:
            ops.bn_running_update(running)
            if pending:
                with ops.ordered("bn_batch_counters"):
                    torch._foreach_add_(pending, 1)
        return False


# Introspection tap for parity tests (the counterpart of a forward hook for what a block does NOT
# materialise): when set to a callable it receives, for every conv of every block evaluation,
# (bn module, raw conv output y, BN scale, BN shift, materialised block output or None) -- exactly the
# tensors the kernels route ReLU / max-pool decisions by.  None (the default) costs nothing.
RAW_TAP = None


def bump_weights_epoch() -> None:
    global _weights_epoch
    _weights_epoch += 1


def packed_weights(w: Tensor, dtype: torch.dtype):
    """(forward image, dgrad image) of a 3x3 weight, cached ON the parameter object until the
    weight changes.  (A global dict keyed by id(w) would hand a new parameter that reuses a dead
    one's id and storage address the dead one's packed weights.)"""
    tag = (w._version, _weights_epoch, w.data_ptr(), torch.cuda.is_current_stream_capturing())
    cache = w.__dict__.get("_cy_pack")
    if cache is None:
        cache = w.__dict__["_cy_pack"] = {}
    hit = cache.get(dtype)
    cur = torch.cuda.current_stream(w.device)
    if hit is not None and hit[0] == tag:
        # packed on another stream (two-stream forward, side-stream wgrad): wait for it -- unless the
        # event belongs to another graph capture (captures are fenced by full synchronisation)
        if hit[4] != cur and hit[5] == ops._capture_id(cur):
            cur.wait_event(hit[3])
        return hit[1], hit[2]
    wf, wd = ops.pack_weights(w, dtype, want_dgrad=True)
    ev = torch.cuda.Event()
    ev.record(cur)
    cache[dtype] = (tag, wf, wd, ev, cur, ops._capture_id(cur))
    return wf, wd


def prepack(weights, dtype: torch.dtype) -> None:
    """pack ALL the given 3x3 weights with one launch if the first one's packed image is stale (they
    change together, with the optimizer step), and file the results where `packed_weights` looks"""
    if not weights:
        return
    w0 = weights[0]
    cap = torch.cuda.is_current_stream_capturing()
    hit = w0.__dict__.get("_cy_pack", {}).get(dtype)
    if hit is not None and hit[0] == (w0._version, _weights_epoch, w0.data_ptr(), cap):
        return
    cur = torch.cuda.current_stream(w0.device)
    packs = ops.pack_weights_batched([w.detach() for w in weights], dtype)
    ev = torch.cuda.Event()
    ev.record(cur)
    capid = ops._capture_id(cur)
    for w, (wf, wd) in zip(weights, packs):
        w.__dict__.setdefault("_cy_pack", {})[dtype] = ((w._version, _weights_epoch, w.data_ptr(), cap),
                                                        wf, wd, ev, cur, capid)


# ---- weight gradients of a layer that two passes of one step go through ------------------------------
# SemiSupervisedEpocher's two-stage step evaluates the network twice (labeled batch; unlabeled batch +
# its transformed view) and both backward passes add into the same .grad.  Instead of two launches
# (each with its own slab traffic and slab reduction) the pass that reaches a weight FIRST in the
# backward order parks its operands, the second one issues ONE launch over both batches
# (cy_conv3x3_wgrad_pair).  A forward-use counter per weight says whether a partner can still come;
# whatever is still parked when the backward pass ends (the partner's branch got no gradient: e.g. the
# decoder of the unlabeled pass) is issued then as an ordinary single launch.
# Backward passes run in reverse forward order (autograd serves the node with the highest sequence
# number first), so the pass that was evaluated FIRST in a step is differentiated last: it never parks
# -- nobody can come after it -- and takes along what the later passes parked.
PAIR_WGRAD = os.environ.get("CY_PAIR_WGRAD", "1") != "0"
_parked = {}  # id(weight) -> _Parked
_keepalive = []  # operands of joint launches issued from another stream, until the backward pass ends
_pass_serial = 0          # network evaluations so far (begin_pass)
_step_first_pass = None   # serial of the first evaluation since the last backward pass ended


def begin_pass() -> int:
    """called by the network's forward: a new evaluation (pass) begins"""
    global _pass_serial, _step_first_pass
    _pass_serial += 1
    if _step_first_pass is None and torch.is_grad_enabled():  # (evaluation passes are not differentiated)
        _step_first_pass = _pass_serial
        if ops.marks_wanted and torch.cuda.is_available():
            ops.begin_step_marks(torch.cuda.current_device())  # (data parallel: a new step's serial)
    return _pass_serial


class _Parked:
    __slots__ = ("w", "sink", "src1", "src2", "dy", "mode", "scale", "shift", "stream", "event", "capid")


def note_forward_use(w: Tensor, needs_grad: bool) -> None:
    """called by a Function's forward (where grad mode is always off: `needs_grad` comes from
    ctx.needs_input_grad, which is False under an outer no_grad)"""
    if PAIR_WGRAD and needs_grad:
        w.__dict__["_cy_uses"] = w.__dict__.get("_cy_uses", 0) + 1


def _flush_parked() -> None:
    global _step_first_pass
    _step_first_pass = None
    _keepalive.clear()
    for key in list(_parked):
        p = _parked.pop(key)
        p.w.__dict__["_cy_uses"] = 0
        with torch.cuda.stream(p.stream):
            ops.conv3x3_wgrad(p.src1, p.src2, p.dy, mode=p.mode, scale=p.scale, shift=p.shift, out=p.sink)
        ops.note_side_work(p.stream)  # (a no-op for the home stream's join; covers a parked side stream)


def wgrad_into_sink(w: Tensor, sink: Tensor, src1: Tensor, src2: Optional[Tensor], dy: Tensor, mode: int,
                    scale: Optional[Tensor], shift: Optional[Tensor], pass_id: int) -> None:
    """sink += dw of one 3x3 conv, pairing the two passes of a step into one launch where it can"""
    uses = w.__dict__.get("_cy_uses", 0)
    if uses > 0:
        # (the step's first pass is differentiated last: whatever the counter still says then -- an
        # evaluation that never reached a loss -- must not leak into the next step)
        uses = 0 if pass_id == _step_first_pass else uses - 1
        w.__dict__["_cy_uses"] = uses
    cur = torch.cuda.current_stream(dy.device)
    p = _parked.pop(id(w), None)
    if p is not None and p.sink is sink and p.mode == mode and p.src1.shape[1:] == src1.shape[1:]:
        if p.stream != cur and p.capid == ops._capture_id(cur):
            cur.wait_event(p.event)
        ops.conv3x3_wgrad_pair(p.src1, p.src2, p.dy, p.scale, p.shift, src1, src2, dy, scale, shift,
                               mode=mode, out=sink)
        if p.stream != cur:
            if ops.CAPTURING:
                # inside a capture record_stream is not available: keep the other stream's operands
                # alive until the backward pass (= the capture) ends, so that its allocator cannot hand
                # their memory out again while this launch reads it
                _keepalive.append(p)
            else:
                for t in (p.src1, p.src2, p.dy, p.scale, p.shift):
                    if t is not None:
                        t.record_stream(cur)
        return
    if p is not None:  # not the same layer geometry after all: issue it on its own
        _parked[id(w)] = p
        _flush_parked()
    if PAIR_WGRAD and uses > 0 and pass_id != _step_first_pass and src1.dtype in ops.HALF_TYPES:
        q = _Parked()
        q.w, q.sink, q.src1, q.src2, q.dy, q.mode, q.scale, q.shift = w, sink, src1, src2, dy, mode, scale, shift
        q.stream, q.event, q.capid = cur, torch.cuda.Event(), ops._capture_id(cur)
        q.event.record(cur)
        _parked[id(w)] = q
        ops.at_backward_end(_flush_parked)
        return
    if _step_first_pass is not None:
        ops.at_backward_end(_flush_parked)  # (also re-arms the pass bookkeeping for the next step)
    ops.conv3x3_wgrad(src1, src2, dy, mode=mode, scale=scale, shift=shift, out=sink)


def compute_dtype_for(x: Tensor, requested: Optional[torch.dtype]) -> torch.dtype:
    """the requested dtype; else, under autocast (AMPScaler.autocast, contrastyou/amp/amp.py:44), the
    autocast dtype -- float16 in the reference's own mode (fp16 + GradScaler), bfloat16 in the build's
    default mode; else the input's half type; else f32 (verification mode)."""
    if requested is not None:
        return requested
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        return dt if dt in ops.HALF_TYPES else torch.bfloat16
    if x.dtype in ops.HALF_TYPES:
        return x.dtype
    return torch.float32


class ChainCfg:
    """Static description of one [conv3x3 -> BN -> ReLU] x {1,2} block."""

    def __init__(self, bns: List[nn.BatchNorm2d], mode: int, first: bool, pool_out: bool = False):
        self.bns = bns
        self.mode = mode  # ops.CY_SRC_* for the first conv's source 1
        self.first = first  # first conv reads the f32 image (Cin <= 4)
        self.pool_out = pool_out  # second output: MaxPool2d(2) of the block output (written by the same launch)
        # data parallel: when this block's backward has run for the last time in a step, the gradients of every
        # parameter carrying this tag or an earlier one are final (ops.grad_ready_mark)
        self.ready_tag: Optional[str] = None
        self.dtype: Optional[torch.dtype] = None  # forced compute dtype (None = infer)


def _bn_flags(bn: nn.BatchNorm2d):
    use_batch = bn.training or bn.running_mean is None
    update = bn.training and bn.track_running_stats and bn.running_mean is not None
    return use_batch, update


class ConvChainFn(torch.autograd.Function):
    """One reference block: _ConvBlock (2 convs) or _UpConv (upsample + 1 conv)."""

    @staticmethod
    def forward(ctx, cfg: ChainCfg, x1: Tensor, x2: Optional[Tensor], *params: Tensor):
        nconv = len(params) // 3
        ops.require_gpu(x1, x2, *params)
        dt = compute_dtype_for(x1, cfg.dtype)
        dev = x1.device
        saved = []
        batch_flags = []
        cur, cur2, mode = x1, x2, cfg.mode
        if not cfg.first:
            if cur.dtype != dt:
                cur = cur.to(dt)
            cur = ops.to_nhwc(cur)
            if cur2 is not None:
                cur2 = ops.to_nhwc(cur2.to(dt) if cur2.dtype != dt else cur2)
        x1s, x2s = cur, cur2
        scale = shift = None
        fold = None  # accumulator path: the BatchNorm evaluation whose coefficients the next kernel derives itself
        ys, coefs, run_items = [], [], []
        for i in range(nconv):
            w, g, b = params[3 * i: 3 * i + 3]
            bn = cfg.bns[i]
            use_batch, update = _bn_flags(bn)
            Cout = w.shape[0]
            acc_i = ops.BN_ACC and use_batch and Cout <= 1024
            if i == 0 and cfg.first:
                y, part = ops.conv_first_fwd(cur, w, dt, want_stats=use_batch, stats_acc=acc_i)
            else:
                note_forward_use(w, ctx.needs_input_grad[3 + 3 * i])
                wf, _ = packed_weights(w, dt)
                y, part = ops.conv3x3_fwd(cur, cur2 if i == 0 else None, wf, Cout,
                                          mode=mode if i == 0 else 0, scale=scale, shift=shift, fold=fold,
                                          want_stats=use_batch, stats_acc=acc_i)
                if fold is not None:
                    fold.done = True
            count = y.shape[0] * y.shape[2] * y.shape[3]
            mom = bn.momentum if bn.momentum is not None else 0.1
            if acc_i:
                # no finalize launch: the consumer of y (the next conv's prologue, or the apply launch below) derives
                # the coefficients from the accumulator and leaves [scale, shift, mean, invstd, var] in fold.coef
                fold = ops.BnState(part, g.detach(), b.detach(), count, bn.eps, dev)
                scale = shift = None
                coefs.append((fold.coef[0], fold.coef[1], fold.coef[2], fold.coef[3]))
                if update:
                    run_items.append((fold.coef, bn.running_mean, bn.running_var, mom))
            else:
                fold = None
                scale, shift, mean, invstd = ops.bn_finalize(
                    part, count, g.detach(), b.detach(), bn.running_mean, bn.running_var, mom, bn.eps, use_batch,
                    update, Cout, dev)
                coefs.append((scale, shift, mean, invstd))
            if update and bn.num_batches_tracked is not None:
                if _nbt_pending is not None:
                    _nbt_pending.append(bn.num_batches_tracked)
                else:
                    bn.num_batches_tracked.add_(1)
            ys.append(y)
            batch_flags.append(use_batch)
            cur, cur2 = y, None
        if fold is not None:
            out, pooled = ops.bn_relu_apply_pool_fold(ys[-1], fold) if cfg.pool_out else (ops.bn_relu_apply_fold(ys[-1], fold), None)
        elif cfg.pool_out:
            out, pooled = ops.bn_relu_apply_pool(ys[-1], scale, shift)
        else:
            out, pooled = ops.bn_relu_apply(ys[-1], scale, shift), None
        if run_items:
            if _run_pending is not None:
                _run_pending.extend(run_items)
            else:
                ops.bn_running_update(run_items)
        # accumulators of the backward sums, from the same zeroed arena (the backward pass allocates nothing to zero)
        ctx.bwd_accs = None
        if ops.BN_ACC and any(ctx.needs_input_grad) and all(yy.shape[1] <= 1024 for yy in ys):
            ctx.bwd_accs = [ops.bn_bwd_acc_new(yy.shape[0], yy.shape[1], yy.shape[2], yy.shape[3],
                                               cfg.pool_out and j == nconv - 1, dev) for j, yy in enumerate(ys)]
            # whoever produces the gradient of `out` may add the sums of this block's last BatchNorm while it has the
            # values in registers (an _UpConv's upsample backward does): what it needs travels with the tensor
            out._cy_tail = (ys[-1], coefs[-1][0], ctx.bwd_accs[-1])
        # ... and this block, if it upsamples on load, is such a producer for the block whose output it reads
        ctx.up_tail = getattr(x1, "_cy_tail", None) if (cfg.mode == ops.CY_SRC_UP2 and ops.BN_ACC) else None
        ctx.x2_tail = getattr(x2, "_cy_tail", None) if (x2 is not None and ops.BN_ACC) else None
        if RAW_TAP is not None:
            for i in range(nconv):
                RAW_TAP(cfg.bns[i], ys[i], coefs[i][0], coefs[i][1], out if i == nconv - 1 else None)
        ctx.cfg, ctx.nconv, ctx.dt = cfg, nconv, dt
        ctx.pass_id = _pass_serial
        ctx.batch_flags = batch_flags
        ctx.x_shape = tuple(x1.shape)
        ctx.x_dtype = x1.dtype
        ctx.has_x2 = x2 is not None
        tensors = [x1s] + ([x2s] if x2 is not None else []) + list(params) + ys
        for c in coefs:
            tensors.extend(c)
        if cfg.pool_out:
            # backward of the pooled output routes through the arg-max of `out`; either output may go unused
            # (a pass whose loss taps only the encoder leaves `out`'s skip branch without a gradient)
            tensors.append(out)
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(*tensors)
            return out, pooled
        ctx.save_for_backward(*tensors)
        return out

    @staticmethod
    def backward(ctx, dout: Optional[Tensor], dpooled: Optional[Tensor] = None):
        ops.ensure_backward_join()
        cfg, nconv, dt = ctx.cfg, ctx.nconv, ctx.dt
        t = list(ctx.saved_tensors)
        out = t.pop() if cfg.pool_out else None
        if cfg.pool_out and dout is None and dpooled is None:
            return (None,) * (3 + 3 * nconv)
        x1 = t.pop(0)
        x2 = t.pop(0) if ctx.has_x2 else None
        params = [t.pop(0) for _ in range(3 * nconv)]
        ys = [t.pop(0) for _ in range(nconv)]
        coefs = [tuple(t.pop(0) for _ in range(4)) for _ in range(nconv)]
        pool_partials = None
        acc_ok = ops.BN_ACC and all(yy.shape[1] <= 1024 for yy in ys)
        accs = ctx.bwd_accs if acc_ok else None
        ctx.bwd_accs = None  # (a second backward through the same node gets fresh accumulators)
        if acc_ok and accs is None:
            accs = [None] * nconv
        pool_acc_filled = False
        if accs is not None and accs[-1] is not None and accs[-1].filled_for is not None and not cfg.pool_out:
            # the producer of `dout` (an upsample backward) has added this block's last BatchNorm's sums -- valid only if
            # `dout` IS that producer's tensor (autograd hands a single contribution through; a sum of several consumers'
            # gradients is another tensor, and the accumulator then holds a partial sum: start a fresh one)
            made = accs[-1].filled_for  # (the producer's tensor, kept alive: no address reuse, no in-place accumulation)
            if (dout is not None and dout.data_ptr() == made.data_ptr() and dout._version == made._version
                    and dout.shape == made.shape and dout.dtype == ys[-1].dtype and ops.is_nhwc(dout)):
                pool_acc_filled = True
            else:
                accs[-1] = None
        if cfg.pool_out and dpooled is not None:
            if dpooled.dtype != out.dtype:
                dpooled = dpooled.to(out.dtype)
            add = None
            if dout is not None:
                add = ops.to_nhwc(dout if dout.dtype == out.dtype else dout.to(out.dtype))
            # the pooled branch's gradient through the arg-max, + the skip branch's gradient; the same launch takes
            # the backward sums of the block's last BatchNorm (whose dA it is writing)
            if accs is not None:
                dout, acc_p = ops.maxpool2_bwd_bn_acc(out, ops.to_nhwc(dpooled), add, ys[-1], coefs[-1][0], accs[-1])
                if acc_p is not None:
                    accs[-1], pool_acc_filled = acc_p, True
            else:
                dout, pool_partials = ops.maxpool2_bwd_bn(out, ops.to_nhwc(dpooled), add, ys[-1], *coefs[-1])
        need = ctx.needs_input_grad  # (cfg, x1, x2, *params)
        grads_p: List[Optional[Tensor]] = [None] * (3 * nconv)
        da = dout
        dx1 = dx2 = None
        dz_filled = False  # the data gradient that produced `da` has added this iteration's BatchNorm's backward sums
        for i in reversed(range(nconv)):
            if i == nconv - 1:
                sums_done = pool_acc_filled
            else:
                sums_done, dz_filled = dz_filled, False
            w, g, b = params[3 * i: 3 * i + 3]
            scale, shift, mean, invstd = coefs[i]
            need_w, need_g, need_b = need[3 + 3 * i], need[3 + 3 * i + 1], need[3 + 3 * i + 2]
            # parameter gradients are accumulated straight into .grad when it is a live f32 buffer
            # (flat-buffer optimizer); otherwise they are returned to autograd as usual
            gsink = ops.grad_sink(g) if (need_g and need_b) else None
            bsink = ops.grad_sink(b) if gsink is not None else None
            if bsink is None:
                gsink = None
            # The data gradient of this conv can take the BatchNorm + ReLU backward into its load path (one launch
            # instead of two; dy comes back as its second output for the weight gradient) where the layer has such a
            # launch plan -- the flow kernel's tilings with room for a second halo buffer.
            Cin_i = w.shape[1]
            want_dgrad = i > 0 or (not cfg.first and (need[1] or (ctx.has_x2 and need[2])))
            split_i = x1.shape[1] if (i == 0 and ctx.has_x2) else None
            fused_dx = None
            if accs is not None and want_dgrad:
                da_f = ops.to_nhwc(da if da.dtype == ys[i].dtype else da.to(ys[i].dtype))
                if ops.conv3x3_dgrad_bn_ok(da_f, Cin_i, split_i):
                    acc_i = accs[i] if accs[i] is not None else ops.bn_bwd_acc_new(*ys[i].shape[:2], *ys[i].shape[2:], False, da_f.device)
                    if not sums_done:
                        ops.bn_bwd_reduce_acc(da_f, ys[i], scale, acc_i)
                    _, wd = packed_weights(w, dt)
                    fused_dx, dy, dgamma, dbeta = ops.conv3x3_dgrad_bn(
                        da_f, ys[i], scale, acc_i, ctx.batch_flags[i], wd, Cin_i, dgamma_out=gsink, dbeta_out=bsink,
                        want_param_grads=need_g or need_b, split=split_i)
            if fused_dx is not None:
                pass
            elif accs is not None:
                # (scale is row 0 of the contiguous coefficient block [scale, shift, mean, invstd, ...] of either path)
                dy, dgamma, dbeta = ops.bn_relu_bwd_acc(da, ys[i], scale, ctx.batch_flags[i], dgamma_out=gsink,
                                                        dbeta_out=bsink, want_param_grads=need_g or need_b,
                                                        acc=accs[i], acc_filled=sums_done)
            else:
                dy, dgamma, dbeta = ops.bn_relu_bwd(da, ys[i], scale, shift, mean, invstd, ctx.batch_flags[i],
                                                    dgamma_out=gsink, dbeta_out=bsink,
                                                    want_param_grads=need_g or need_b,
                                                    partials=pool_partials if i == nconv - 1 else None)
            if need_g:
                grads_p[3 * i + 1] = dgamma
            if need_b:
                grads_p[3 * i + 2] = dbeta
            wsink = ops.grad_sink(w) if need_w else None
            if i > 0:
                ps, ph = coefs[i - 1][0], coefs[i - 1][1]
                if need_w:
                    if wsink is not None and ops.ASYNC_WGRAD:
                        with ops.on_side_stream(ys[i - 1], dy, ps, ph):
                            wgrad_into_sink(w, wsink, ys[i - 1], None, dy, 0, ps, ph, ctx.pass_id)
                    elif wsink is not None:
                        wgrad_into_sink(w, wsink, ys[i - 1], None, dy, 0, ps, ph, ctx.pass_id)
                    else:
                        grads_p[3 * i] = ops.conv3x3_wgrad(ys[i - 1], None, dy, scale=ps, shift=ph)
                if fused_dx is not None:
                    da = fused_dx
                else:
                    _, wd = packed_weights(w, dt)
                    # the output is the dA of this block's previous BatchNorm: its backward sums from the epilogue
                    # where the launch plan allows (then the next iteration skips the reduce launch)
                    if (accs is not None and accs[i - 1] is not None
                            and ops.conv3x3_dgrad_dz_ok(dy, w.shape[1], None, 0, w.shape[1])):
                        da = ops.conv3x3_dgrad_dz(dy, wd, w.shape[1], ys[i - 1], coefs[i - 1][0], accs[i - 1])
                        dz_filled = True
                    else:
                        da, _ = ops.conv3x3_fwd(dy, None, wd, w.shape[1], want_stats=False)
            else:
                if need_w:
                    if wsink is not None and ops.ASYNC_WGRAD:
                        with ops.on_side_stream(x1, x2, dy):
                            if cfg.first:
                                ops.conv_first_wgrad(x1, dy, out=wsink)
                            else:
                                wgrad_into_sink(w, wsink, x1, x2, dy, cfg.mode, None, None, ctx.pass_id)
                    else:
                        if cfg.first:
                            dw = ops.conv_first_wgrad(x1, dy, out=wsink)
                        elif wsink is not None:
                            wgrad_into_sink(w, wsink, x1, x2, dy, cfg.mode, None, None, ctx.pass_id)
                            dw = None
                        else:
                            dw = ops.conv3x3_wgrad(x1, x2, dy, mode=cfg.mode)
                        grads_p[0] = None if wsink is not None else dw
                need_x1 = need[1]
                need_x2 = ctx.has_x2 and need[2]
                if need_x1 or need_x2:
                    if cfg.first:
                        raise RuntimeError("gradient w.r.t. the input image is not implemented "
                                           "(the reference never asks for it)")
                    _, wd = packed_weights(w, dt)
                    C1 = x1.shape[1]
                    if fused_dx is not None:
                        if ctx.has_x2:
                            dl1, dl2 = fused_dx
                            dx2 = dl2 if need_x2 else None
                        else:
                            dl1 = fused_dx
                    elif ctx.has_x2:
                        t2 = ctx.x2_tail
                        if (t2 is not None and need_x2 and t2[2].filled_for is None
                                and ops.conv3x3_dgrad_dz_ok(dy, w.shape[1], C1, C1, w.shape[1] - C1)):
                            # the second part of the output is the dA of the block that produced x2 (an _UpConv): its
                            # BatchNorm's backward sums from this launch's epilogue
                            dl1, dl2 = ops.conv3x3_dgrad_dz(dy, wd, w.shape[1], t2[0], t2[1], t2[2], split=C1)
                            t2[2].filled_for = dl2
                        else:
                            (dl1, dl2), _ = ops.conv3x3_fwd(dy, None, wd, w.shape[1], want_stats=False,
                                                            split=C1)
                        dx2 = dl2 if need_x2 else None
                    else:
                        dl1, _ = ops.conv3x3_fwd(dy, None, wd, w.shape[1], want_stats=False)
                    if need_x1:
                        if cfg.mode == ops.CY_SRC_POOL2:
                            dx1 = ops.maxpool2_bwd(x1, dl1)
                        elif cfg.mode == ops.CY_SRC_UP2:
                            dx1 = None
                            if ctx.up_tail is not None and ctx.up_tail[2].filled_for is None:
                                y_t, coef_t, acc_t = ctx.up_tail
                                dx1 = ops.upsample2_bwd_bn_acc(dl1, y_t, coef_t, acc_t)
                                if dx1 is not None:
                                    acc_t.filled_for = dx1
                            if dx1 is None:
                                dx1 = ops.upsample2_bwd(dl1)
                        else:
                            dx1 = dl1
                        if dx1.dtype != ctx.x_dtype:
                            dx1 = dx1.to(ctx.x_dtype)
        if cfg.ready_tag is not None and (_step_first_pass is None or ctx.pass_id == _step_first_pass):
            # (the pass evaluated first is differentiated last: nothing of this block is parked or still to come)
            ops.grad_ready_mark(cfg.ready_tag, da.device)
        return (None, dx1, dx2, *grads_p)


class HeadFn(torch.autograd.Function):
    """nn.Conv2d(C, K, 1) with bias (contrastyou/arch/unet.py:102); f32 logits."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor]):
        ops.require_gpu(x, w)
        x = ops.to_nhwc(x)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.bias = b  # the parameter object (for its .grad sink), not saved data
        return ops.head_fwd(x, w, b)

    @staticmethod
    def backward(ctx, dlogits: Tensor):
        x, w = ctx.saved_tensors
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (
            ctx.has_bias and ctx.needs_input_grad[2])
        # like the conv / BN parameters: the reduction kernel adds straight into live .grad buffers (ordered across
        # streams), so that autograd never has to merge the two passes' contributions itself
        wsink = ops.grad_sink(w) if (need_dw and ctx.needs_input_grad[1] and w.dtype == torch.float32) else None
        bsink = ops.grad_sink(ctx.bias) if (wsink is not None and ctx.has_bias and ctx.needs_input_grad[2]) else None
        if wsink is not None and (bsink is not None or not ctx.has_bias):
            dx, _, _ = ops.head_bwd(x, w, dlogits, need_dx, True, dw_into=wsink, db_into=bsink)
            return dx, None, None
        dx, dw, db = ops.head_bwd(x, w, dlogits, need_dx, need_dw)
        if dw is not None:
            dw = dw.to(w.dtype)
        return dx, dw if ctx.needs_input_grad[1] else None, db if ctx.has_bias else None


class SoftmaxKLFn(torch.autograd.Function):
    """KL_div(softmax(logits,1), one_hot(target)) with mean reduction
    (semi_seg/epochers/epocher.py:317-318, contrastyou/losses/kl.py:112-125)."""

    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, eps: float):
        ops.require_gpu(logits, target)
        logits = ops.to_nhwc(logits.float())
        target = target.contiguous()
        ctx.save_for_backward(logits, target)
        ctx.eps = eps
        return ops.softmax_kl_fwd(logits, target, eps)

    @staticmethod
    def backward(ctx, g: Tensor):
        logits, target = ctx.saved_tensors
        gs = g.reshape(1).float().contiguous()
        return ops.softmax_kl_bwd(logits, target, gs, ctx.eps), None, None


class SoftmaxMSEFn(torch.autograd.Function):
    """nn.MSELoss()(a.softmax(1), b.softmax(1)) (semi_seg/hooks/consistency.py:36, mt.py:186)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        ops.require_gpu(a, b)
        a = ops.to_nhwc(a.float())
        b = ops.to_nhwc(b.float())
        ctx.save_for_backward(a, b)
        return ops.softmax_mse_fwd(a, b)

    @staticmethod
    def backward(ctx, g: Tensor):
        a, b = ctx.saved_tensors
        gs = g.reshape(1).float().contiguous()
        da, db = ops.softmax_mse_bwd(a, b, gs, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return da, db


class AvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((1,1)) + Flatten (contrastyou/projectors/heads.py:15-16)."""

    @staticmethod
    def forward(ctx, x: Tensor):
        ops.require_gpu(x)
        x = ops.to_nhwc(x)
        ctx.shape, ctx.dtype = tuple(x.shape), x.dtype
        return ops.avgpool_fwd(x)

    @staticmethod
    def backward(ctx, g: Tensor):
        return ops.avgpool_bwd(g.float().contiguous(), ctx.shape, ctx.dtype)


class LinearFn(torch.autograd.Function):
    """nn.Linear (+ optional in-place LeakyReLU), contrastyou/projectors/heads.py:17-19."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], act: int, slope: float):
        ops.require_gpu(x, w)
        x = x.float().contiguous()
        w32 = w.detach().float().contiguous()
        y = ops.linear_fwd(x, w32, None if b is None else b.detach().float().contiguous(), act, slope)
        ctx.save_for_backward(x, w32, y)
        ctx.act, ctx.slope, ctx.has_bias = act, slope, b is not None
        ctx.params = (w, b)  # (leaf parameters: their .grad buffers take the gradients directly when they can)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        x, w, y = ctx.saved_tensors
        need_dx = ctx.needs_input_grad[0]
        need_dw = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        wp, bp = ctx.params
        wsink = bsink = None
        if ctx.needs_input_grad[1] and (not ctx.has_bias or ctx.needs_input_grad[2]):
            wsink = ops.grad_sink(wp)
            bsink = ops.grad_sink(bp) if ctx.has_bias and wsink is not None else None
            if ctx.has_bias and bsink is None:
                wsink = None
        dx, dw, db = ops.linear_bwd(x, w, y, g.float().contiguous(), ctx.act, ctx.slope, need_dx,
                                    need_dw, dw_into=wsink, db_into=bsink)
        return dx, dw if ctx.needs_input_grad[1] else None, db if ctx.has_bias else None, None, None


class ProjHeadFn(torch.autograd.Function):
    """ProjectionHead, head_type "mlp" with normalisation, as ONE launch forward and two backward
    (contrastyou/projectors/heads.py:12-22,81-96): pool -> Linear -> LeakyReLU(0.01) -> Linear -> F.normalize."""

    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor):
        ops.require_gpu(x, w1, w2)
        x = ops.to_nhwc(x)
        z, pooled, y1, y2, norms = ops.proj_head_fwd(x, w1.detach().float().contiguous(), b1.detach().float().contiguous(),
                                                     w2.detach().float().contiguous(), b2.detach().float().contiguous())
        ctx.save_for_backward(pooled, y1, y2, norms, w1.detach(), w2.detach())
        ctx.shape, ctx.dtype = tuple(x.shape), x.dtype
        ctx.params = (w1, b1, w2, b2)
        return z

    @staticmethod
    def backward(ctx, g: Tensor):
        pooled, y1, y2, norms, w1, w2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        sinks = None
        if all(need[1:5]):
            cand = tuple(ops.grad_sink(p) for p in ctx.params)
            if all(c is not None for c in cand):
                sinks = cand
        dx, (dw1, db1, dw2, db2) = ops.proj_head_bwd(g.float().contiguous(), pooled, y1, y2, norms,
                                                     w1.float().contiguous(), w2.float().contiguous(), ctx.shape,
                                                     ctx.dtype, need[0], sinks)
        return (dx, dw1 if need[1] else None, db1 if need[2] else None, dw2 if need[3] else None,
                db2 if need[4] else None)


class L2NormFn(torch.autograd.Function):
    """F.normalize(x, p=2, dim=1) on [M,D] (contrastyou/projectors/nn.py:47-54)."""

    @staticmethod
    def forward(ctx, x: Tensor):
        ops.require_gpu(x)
        x = x.float().contiguous()
        z, norms = ops.l2norm_fwd(x)
        ctx.save_for_backward(x, norms)
        return z

    @staticmethod
    def backward(ctx, g: Tensor):
        x, norms = ctx.saved_tensors
        return ops.l2norm_bwd(x, norms, g.float().contiguous())


class SupConFn(torch.autograd.Function):
    """SupConLoss1._forward (contrastyou/losses/contrastive.py:52-100) on P = cat(z1, z2).  Returns the loss, the
    diagonal of the similarity matrix (|P_i|^2 / t: what the reference's unit-norm assertion inspects) and the row
    statistics; for D <= 256 the 2n x 2n matrix itself exists only tile-wise inside the kernels."""

    @staticmethod
    def forward(ctx, P: Tensor, labels: Optional[Tensor], pos_mask: Optional[Tensor], t: float, exclude_pos: bool = False):
        ops.require_gpu(P)
        P = P.float().contiguous()
        ctx.excl = bool(exclude_pos)
        ctx.fused = ops.supcon_fused_ok(P) and not ctx.excl
        if ctx.excl:  # exclude_other_pos=True (contrastive.py:87-91): on the materialised matrix
            loss, S, stats, tmp = ops.supcon_excl_fwd(P, labels, pos_mask, t)
            diag = S.diagonal().clone()
            ctx.save_for_backward(P, stats, S, tmp)
        elif ctx.fused:
            loss, diag, stats = ops.supcon_fwd_fused(P, labels, pos_mask, t)
            ctx.save_for_backward(P, stats)
        else:
            loss, S, stats = ops.supcon_fwd(P, labels, pos_mask, t)
            diag = S.diagonal().clone()
            ctx.save_for_backward(P, stats, S)
        ctx.labels, ctx.pos_mask, ctx.t = labels, pos_mask, t
        ctx.mark_non_differentiable(diag, stats)
        return loss, diag, stats

    @staticmethod
    def backward(ctx, g: Tensor, _gd, _gstats):
        gs = g.reshape(1).float().contiguous()
        if ctx.excl:
            P, stats, S, tmp = ctx.saved_tensors
            return ops.supcon_excl_bwd(P, ctx.labels, ctx.pos_mask, S, stats, tmp, gs, ctx.t), None, None, None, None
        if ctx.fused:
            P, stats = ctx.saved_tensors
            return ops.supcon_bwd_fused(P, ctx.labels, ctx.pos_mask, stats, gs, ctx.t), None, None, None, None
        P, stats, S = ctx.saved_tensors
        return ops.supcon_bwd(P, ctx.labels, ctx.pos_mask, S, stats, gs, ctx.t), None, None, None, None


class AffineFn(torch.autograd.Function):
    """Nearest-neighbour affine resampling (+gamma), the in-step augmentation of
    semi_seg/augment.py:297-311 with the geometry given explicitly as theta."""

    @staticmethod
    def forward(ctx, x: Tensor, theta: Tensor, gamma: Optional[Tensor]):
        ops.require_gpu(x, theta)
        ctx.save_for_backward(theta)
        ctx.has_gamma = gamma is not None
        return ops.affine_fwd(x, theta, gamma)

    @staticmethod
    def backward(ctx, g: Tensor):
        if ctx.has_gamma:
            raise RuntimeError("backward through the gamma (image-mode) transform is not defined: "
                               "the reference applies it to input images only")
        (theta,) = ctx.saved_tensors
        return ops.affine_bwd(g, theta), None, None


# ------------------------------------------------------------------------------------------------
# dense contrastive projector, point sampling, cluster heads, discrete MI, GroupNorm block
# ------------------------------------------------------------------------------------------------
class DenseProjHiddenFn(torch.autograd.Function):
    """mean over adaptive-pool bins of LeakyReLU(Conv1x1(x)) -> [bins, hid]: the fused front half of
    DenseProjectionHead (contrastyou/projectors/heads.py:31-41,99-123); the second 1x1 conv commutes
    with the average pool and is applied to the pooled rows."""

    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, b1: Tensor, size, bins: Optional[Tensor]):
        ops.require_gpu(x, w1)
        x = ops.to_nhwc(x)
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            x = x.float()
        w = w1.detach().reshape(w1.shape[0], -1).float().contiguous()
        b = b1.detach().float().contiguous()
        ctx.save_for_backward(x, w, b)
        ctx.size, ctx.bins = tuple(size), bins
        return ops.dense_proj_fwd(x, w, b, ctx.size, bins)

    @staticmethod
    def backward(ctx, g: Tensor):
        x, w, b = ctx.saved_tensors
        need_dx = ctx.needs_input_grad[0]
        need_dw = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dw, db = ops.dense_proj_bwd(x, w, b, ctx.size, ctx.bins, g.float().contiguous(), need_dx, need_dw)
        if dw is not None:
            dw = dw.view(dw.shape[0], dw.shape[1], 1, 1)
        return dx, dw if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None, None, None


class AdaptiveAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(size) on an NHWC map -> f32 rows [N*sh*sw, C]"""

    @staticmethod
    def forward(ctx, x: Tensor, size):
        ops.require_gpu(x)
        x = ops.to_nhwc(x)
        ctx.shape, ctx.dtype, ctx.size = tuple(x.shape), x.dtype, tuple(size)
        return ops.adaptive_avgpool_fwd(x, ctx.size)

    @staticmethod
    def backward(ctx, g: Tensor):
        return ops.adaptive_avgpool_bwd(g.float().contiguous(), ctx.shape, ctx.dtype, ctx.size), None


class AdaptiveMaxPoolFn(torch.autograd.Function):
    """nn.AdaptiveMaxPool2d(size) on an NHWC map -> f32 rows [N*sh*sw, C] (contrastyou/projectors/nn.py:16-23,
    pool_name="adaptive_max")"""

    @staticmethod
    def forward(ctx, x: Tensor, size):
        ops.require_gpu(x)
        x = ops.to_nhwc(x)
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            x = x.float()
        ctx.shape, ctx.dtype, ctx.size = tuple(x.shape), x.dtype, tuple(size)
        out, arg = ops.adaptive_maxpool_fwd(x, ctx.size)
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        (arg,) = ctx.saved_tensors
        return ops.adaptive_maxpool_bwd(g.float().contiguous(), arg, ctx.shape, ctx.dtype, ctx.size), None


class GatherRowsFn(torch.autograd.Function):
    """rows[idx] for distinct idx (the point sampling of semi_seg/hooks/infonce.py:31-46)"""

    @staticmethod
    def forward(ctx, src: Tensor, idx: Tensor):
        src = src.float().contiguous()
        ctx.save_for_backward(idx)
        ctx.rows = src.shape[0]
        return ops.gather_rows_fwd(src, idx)

    @staticmethod
    def backward(ctx, g: Tensor):
        (idx,) = ctx.saved_tensors
        return ops.gather_rows_bwd(g.float().contiguous(), idx, ctx.rows), None


class GroupSoftmaxFn(torch.autograd.Function):
    """[M, S*k] logits -> [S, M, k] probabilities, softmax(logits / T) inside each sub-head
    (SoftmaxWithT, contrastyou/projectors/nn.py:35-44)"""

    @staticmethod
    def forward(ctx, logits: Tensor, S: int, k: int, T: float):
        logits = logits.float().contiguous()
        probs = ops.group_softmax_fwd(logits, S, k, T)
        ctx.save_for_backward(probs)
        ctx.T = T
        return probs

    @staticmethod
    def backward(ctx, g: Tensor):
        (probs,) = ctx.saved_tensors
        return ops.group_softmax_bwd(probs, g.float().contiguous(), ctx.T), None, None, None


class ClusterHeadFn(torch.autograd.Function):
    """stacked 1x1 conv / linear (S*k <= 128 outputs) + per-sub-head softmax(./T) in one pass on the f32 MFMA:
    rows x [M, C] -> probs [S, M, k]; the logits never reach memory (csrc/cy_cluster_head.hip;
    contrastyou/projectors/heads.py:125-173, projectors/nn.py:35-44)"""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], S: int, k: int, T: float):
        ops.require_gpu(x, w)
        x = x.contiguous()
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            x = x.float()
        w2 = w.detach().reshape(S * k, -1).float().contiguous()
        probs = ops.cluster_head_fwd(x, w2, None if b is None else b.detach().float().contiguous(), S, k, T)
        ctx.save_for_backward(x, w2, probs)
        ctx.T, ctx.w_shape, ctx.has_bias = T, w.shape, b is not None
        return probs

    @staticmethod
    def backward(ctx, g: Tensor):
        x, w2, probs = ctx.saved_tensors
        need_dx = ctx.needs_input_grad[0]
        need_dw = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dx, dw, db = ops.cluster_head_bwd(x, w2, probs, g.float().contiguous(), ctx.T, need_dx, need_dw)
        return (dx, dw.view(ctx.w_shape) if (dw is not None and ctx.needs_input_grad[1]) else None,
                db if (ctx.has_bias and ctx.needs_input_grad[2]) else None, None, None, None)


class IIDFn(torch.autograd.Function):
    """joint of two probability maps + information loss in one autograd node.
    x1, x2: f32 [N,H,W,k] contiguous (vectors: H=W=1).  mode 0/1: IIDSegmentationLoss with padding
    0 / >0; mode 2: IIDLoss (contrastyou/losses/discreteMI.py:90-170,201-261).
    Returns (loss, loss with lambda=1, normalised joint [T*T,k,k])."""

    @staticmethod
    def forward(ctx, x1: Tensor, x2: Tensor, mode: int, pad: int, symmetric: bool, lamda: float, eps: float):
        ops.require_gpu(x1, x2)
        N, H, W, k = x1.shape
        normalise = mode == 0
        J = ops.joint_fwd(x1, x2, N, H, W, k, pad, normalise)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        out2, P, dJ = ops.iid_loss(J, mode, symmetric, lamda, eps, want_grad=need)
        ctx.save_for_backward(x1, x2, dJ)
        ctx.cfg = (N, H, W, k, pad, normalise)
        ctx.mark_non_differentiable(P)
        return out2[0], out2[1].detach(), P

    @staticmethod
    def backward(ctx, g: Tensor, _g1, _gP):
        x1, x2, dJ = ctx.saved_tensors
        N, H, W, k, pad, normalise = ctx.cfg
        gs = g.reshape(1).float().contiguous()
        d1, d2 = ops.joint_bwd(x1, x2, dJ, gs, N, H, W, k, pad, normalise, ctx.needs_input_grad[0],
                               ctx.needs_input_grad[1])
        return d1, d2, None, None, None, None, None


class JointFn(torch.autograd.Function):
    """the k x k joint of two probability maps, (1/npix) sum_p x1[p,:] x2[p,:]^T, as its own autograd node
    (compute_joint_2D_with_padding_zeros, contrastyou/losses/discreteMI.py:246-261, before symmetrisation):
    the pixel contraction on the HIP joint kernels, whatever follows on the k x k result in plain autograd.
    x1, x2: f32 [N,H,W,k] contiguous."""

    @staticmethod
    def forward(ctx, x1: Tensor, x2: Tensor):
        ops.require_gpu(x1, x2)
        N, H, W, k = x1.shape
        ctx.save_for_backward(x1, x2)
        return ops.joint_fwd(x1, x2, N, H, W, k, 0, True).view(k, k)

    @staticmethod
    def backward(ctx, g: Tensor):
        x1, x2 = ctx.saved_tensors
        N, H, W, k = x1.shape
        one = torch.ones(1, device=g.device, dtype=torch.float32)
        d1, d2 = ops.joint_bwd(x1, x2, g.float().contiguous().view(1, k, k), one, N, H, W, k, 0, True,
                               ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return d1, d2


class SgemmFn(torch.autograd.Function):
    """alpha * A @ B^T on the exact f32 MFMA kernel, differentiable in both operands"""

    @staticmethod
    def forward(ctx, A: Tensor, B: Tensor, alpha: float):
        ops.require_gpu(A, B)
        A, B = A.float().contiguous(), B.float().contiguous()
        ctx.save_for_backward(A, B)
        ctx.alpha = alpha
        return ops.sgemm(A, B, alpha, True)

    @staticmethod
    def backward(ctx, g: Tensor):
        A, B = ctx.saved_tensors
        g = g.float().contiguous()
        dA = ops.sgemm(g, B, ctx.alpha, False) if ctx.needs_input_grad[0] else None
        dB = ops.sgemm(g.t().contiguous(), A, ctx.alpha, False) if ctx.needs_input_grad[1] else None
        return dA, dB, None


class GNSiLUFn(torch.autograd.Function):
    """GroupNorm(G, C)(y + bias) -> SiLU on a raw (bias-free) conv output (arch/unet2.py:208-224)"""

    @staticmethod
    def forward(ctx, y: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, groups: int, eps: float):
        y = ops.to_nhwc(y)
        b = None if bias is None else bias.detach().float().contiguous()
        gm, bt = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        out, mr = ops.gn_silu_fwd(y, b, gm, bt, groups, eps)
        ctx.save_for_backward(y, b, gm, bt, mr)
        ctx.groups = groups
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        y, b, gm, bt, mr = ctx.saved_tensors
        g = ops.to_nhwc(g if g.dtype == y.dtype else g.to(y.dtype))
        du, dg, dbt, dbias = ops.gn_silu_bwd(y, g, b, gm, bt, mr, ctx.groups)
        return du, dbias, dg, dbt, None, None


class GNSiLUModFn(torch.autograd.Function):
    """GroupNorm(G, C)(y + bias) * (scale + 1) + shift -> SiLU: `Block.forward(x, scale_shift)` of the reference's
    time-embedded ResnetBlock (arch/unet2.py:208-224,240-246); scale / shift are [N, C] (f32)"""

    @staticmethod
    def forward(ctx, y: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, scale: Tensor, shift: Tensor,
                groups: int, eps: float):
        y = ops.to_nhwc(y)
        b = None if bias is None else bias.detach().float().contiguous()
        gm, bt = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        ms, mt = scale.detach().float().contiguous(), shift.detach().float().contiguous()
        out, mr = ops.gn_silu_mod_fwd(y, b, gm, bt, ms, mt, groups, eps)
        ctx.save_for_backward(y, b, gm, bt, ms, mt, mr)
        ctx.groups = groups
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        y, b, gm, bt, ms, mt, mr = ctx.saved_tensors
        g = ops.to_nhwc(g if g.dtype == y.dtype else g.to(y.dtype))
        du, dg, dbt, dbias, dms, dmt = ops.gn_silu_mod_bwd(y, g, b, gm, bt, ms, mt, mr, ctx.groups)
        return du, dbias, dg, dbt, dms, dmt, None, None


class ActFn(torch.autograd.Function):
    """elementwise SiLU (kind 0) / exact GELU (kind 1) on a small f32 tensor (UNet2's time-embedding MLPs)"""

    @staticmethod
    def forward(ctx, x: Tensor, kind: int):
        x = x.float().contiguous()
        ctx.save_for_backward(x)
        ctx.kind = kind
        return ops.act_fwd(x, kind)

    @staticmethod
    def backward(ctx, g: Tensor):
        (x,) = ctx.saved_tensors
        return ops.act_bwd(x, g.float().contiguous(), ctx.kind), None


class Conv3x3Fn(torch.autograd.Function):
    """bias-free 3x3 convolution (stride 1, padding 1) on the implicit-GEMM kernels; NHWC in/out.
    Used by the GroupNorm block, whose conv bias is folded into the normalisation kernels."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor):
        ops.require_gpu(x, w)
        x = ops.to_nhwc(x)
        wf, wd = packed_weights(w, x.dtype)
        out, _ = ops.conv3x3_fwd(x, None, wf, w.shape[0], want_stats=False)
        ctx.save_for_backward(x, w)
        ctx.wd = wd
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        x, w = ctx.saved_tensors
        g = ops.to_nhwc(g if g.dtype == x.dtype else g.to(x.dtype))
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx, _ = ops.conv3x3_fwd(g, None, ctx.wd, x.shape[1], want_stats=False)
        if ctx.needs_input_grad[1]:
            dw = ops.conv3x3_wgrad(x, None, g).to(w.dtype)
        return dx, dw


def bilinear_resize(x: Tensor, size) -> Tensor:
    """F.interpolate(x, size=size, mode="bilinear") (align_corners=False) for inputs that carry no
    gradient -- the reference resizes input images only (semi_seg/hooks/cc.py:132)"""
    if x.requires_grad:
        raise RuntimeError("bilinear_resize: the HIP kernel is forward-only (input images carry no gradient)")
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        x = x.float()
    return ops.bilinear_fwd(x, tuple(size))
