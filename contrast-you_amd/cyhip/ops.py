"""Tensor-level wrappers over the C ABI (one Python function per kernel family).

torch is used for device memory (caching allocator), the current HIP stream and nothing
else: every function below enqueues hand-written gfx950 kernels from libcontrastyou_hip.so.
Activations are logical [N,C,H,W] tensors in channels_last memory (= NHWC).
"""
from __future__ import annotations

import contextlib
import os
import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import CY_BF16, CY_F16, CY_F32, CY_SRC_DIRECT, CY_SRC_POOL2, CY_SRC_UP2, ConvDesc  # noqa: F401

_DT = {torch.float32: CY_F32, torch.bfloat16: CY_BF16, torch.float16: CY_F16}
HALF_TYPES = (torch.bfloat16, torch.float16)

# bench.py instrumentation: when a list, every conv launch appends (family, algorithmic FLOPs,
# start event, end event); events are recorded on the launch stream (torch's current stream).
PROFILE = None


class TimingEvent:
    """HIP event for measuring only (hipEventDisableSystemFence): torch.cuda.Event creates default events, whose
    recording performs a system-scope release -- an L2 writeback + invalidate that, around a single launch, falls
    inside the measured interval and slows the following kernel.  Recorded on torch's current stream."""

    __slots__ = ("_h",)

    def __init__(self):
        h = C.c_void_p()
        _lib.call("cy_debug_event_create", C.byref(h))
        self._h = h

    def record(self):
        _lib.call("cy_debug_event_record", self._h, _stream())
        return self

    def elapsed_time(self, other: "TimingEvent") -> float:
        """milliseconds from this event to `other` (waits for `other`) -- torch.cuda.Event's signature"""
        us = C.c_float()
        _lib.call("cy_debug_event_elapsed_us", self._h, other._h, C.byref(us))
        return us.value * 1e-3

    def __del__(self):
        try:
            _lib.call("cy_debug_event_destroy", self._h)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def _prof_begin():
    if PROFILE is None:
        return None
    return TimingEvent().record()


def _prof_end(e0, kind: str, flops: float, nbytes: float = 0.0):
    if e0 is None:
        return
    PROFILE.append((kind, flops, e0, TimingEvent().record(), nbytes))


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}: the HIP path computes in float32, bfloat16 or float16") from None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """raw hipStream_t of torch's current stream on the current device"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class PinnedRing:
    """small ring of pinned host buffers for sync-free H2D copies of per-step parameters"""

    def __init__(self, slots: int = 16):
        self._slots, self._bufs, self._i = slots, {}, 0

    def upload(self, host: Tensor, device) -> Tensor:
        key = (tuple(host.shape), host.dtype)
        ring = self._bufs.get(key)
        if ring is None:
            ring = self._bufs[key] = [torch.empty(host.shape, dtype=host.dtype).pin_memory()
                                      for _ in range(self._slots)]
        self._i = (self._i + 1) % self._slots
        buf = ring[self._i]
        buf.copy_(host)
        return buf.to(device, non_blocking=True)


pinned = PinnedRing()


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "contrast-you_amd: the hot path only runs as hand-written HIP kernels on a GPU "
                f"(got a {t.device} tensor); there is no CPU fallback")


def is_nhwc(t: Tensor) -> bool:
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def to_nhwc(t: Tensor) -> Tensor:
    """Return t with NHWC memory (no copy when it already is)."""
    if is_nhwc(t):
        return t
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def empty_nhwc(N: int, Cc: int, H: int, W: int, dtype, device) -> Tensor:
    return torch.empty((N, H, W, Cc), dtype=dtype, device=device).permute(0, 3, 1, 2)


def _f32(n, device) -> Tensor:
    return torch.empty(n, dtype=torch.float32, device=device)


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------- conv3x3
def packed_dims(Cout: int, Cin: int) -> Tuple[int, int]:
    a, b = C.c_int(), C.c_int()
    _lib.call("cy_conv3x3_packed_dims", Cout, Cin, C.byref(a), C.byref(b))
    return a.value, b.value


def packed_elems(Cout: int, Cin: int, dtype: torch.dtype) -> int:
    """elements of a packed weight image: [9, co_pad, ci_pad], then (16-bit, Cout % 64 == 0, Cin % 16 == 0) the
    stage-contiguous image of the LDS-DMA kernel (csrc/cy_conv_flow.h)"""
    return int(_lib.load().cy_conv3x3_packed_elems(Cout, Cin, dtype_code(dtype)))


def pack_weights(w: Tensor, dtype: torch.dtype, want_dgrad: bool = True):
    """w: [Cout,Cin,3,3] f32 -> (wf, wd or None): flat packed images of `dtype` (forward GEMM / data-gradient
    GEMM), packed_elems(Cout, Cin) / packed_elems(Cin, Cout) elements."""
    require_gpu(w)
    Cout, Cin = w.shape[0], w.shape[1]
    w = w.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    wf = torch.empty(packed_elems(Cout, Cin, dtype), dtype=dtype, device=w.device)
    wd = None
    if want_dgrad:
        wd = torch.empty(packed_elems(Cin, Cout, dtype), dtype=dtype, device=w.device)
    _lib.call("cy_conv3x3_pack_weights", w.data_ptr(), wf.data_ptr(), _ptr(wd), Cout, Cin,
              dtype_code(dtype), _stream())
    return wf, wd


class _PackSet:
    """device table + arena layout for packing a fixed list of 3x3 weights in one launch"""

    def __init__(self, weights, dtype):
        items = (_lib.PackItem * len(weights))()
        self.views = []  # per weight: (offset, elements) of the forward and of the dgrad image
        off_f = off_d = first = 0
        for it, w in zip(items, weights):
            Cout, Cin = w.shape[0], w.shape[1]
            cop, cip = packed_dims(Cout, Cin)
            cip2, cop2 = packed_dims(Cin, Cout)
            nf, nd = packed_elems(Cout, Cin, dtype), packed_elems(Cin, Cout, dtype)
            it.w, it.off_f, it.off_d, it.first = w.data_ptr(), off_f, off_d, first
            it.Cout, it.Cin, it.co_pad, it.ci_pad, it.ci_pad2, it.co_pad2 = Cout, Cin, cop, cip, cip2, cop2
            it.off_ff = off_f + 9 * cop * cip if nf > 9 * cop * cip else -1
            it.off_fd = off_d + 9 * cip2 * cop2 if nd > 9 * cip2 * cop2 else -1
            self.views.append(((off_f, nf), (off_d, nd)))
            off_f += nf
            off_d += nd
            first += -(-max(cop, cop2) // 32) * -(-max(cip, cip2) // 32)  # 32 x 32 (co, ci) tiles
        self.n, self.size_f, self.size_d, self.total = len(weights), off_f, off_d, first
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        self.table = raw.to(weights[0].device)


_pack_sets = {}


def pack_weights_batched(weights, dtype: torch.dtype):
    """[(wf, wd)] for a list of [Cout,Cin,3,3] f32 weights, packed by ONE kernel launch into two fresh
    arenas (the per-layer views alias them).  The layer table lives on the device and is built once per
    (weight storages, dtype)."""
    require_gpu(*weights)
    for w in weights:
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise ValueError("pack_weights_batched wants contiguous f32 weights")
    if len(weights) > 64:
        raise ValueError("at most 64 layers per batched pack")
    key = (dtype, tuple(w.data_ptr() for w in weights), tuple(tuple(w.shape) for w in weights))
    ps = _pack_sets.get(key)
    if ps is None:
        if len(_pack_sets) > 8:
            _pack_sets.clear()
        ps = _pack_sets[key] = _PackSet(weights, dtype)
    dev = weights[0].device
    af = torch.empty(ps.size_f, dtype=dtype, device=dev)
    ad = torch.empty(ps.size_d, dtype=dtype, device=dev)
    _lib.call("cy_conv3x3_pack_weights_batched", ps.table.data_ptr(), ps.n, ps.total, af.data_ptr(),
              ad.data_ptr(), dtype_code(dtype), _stream())
    out = []
    for (of, nf), (od, nd) in ps.views:
        out.append((af[of:of + nf], ad[od:od + nd]))
    return out


_desc_cache = {}


class _Plan:
    """a conv descriptor with the plan numbers the library derives from it (queried once per shape:
    the host issue path is the step's critical resource)"""
    __slots__ = ("d", "ref", "npart", "fwd_ws", "wgrad_ws", "pair_ws", "stat_wgs", "dgrad_bn")

    def __init__(self, d: ConvDesc):
        self.d, self.ref = d, C.byref(d)
        self.npart = self.fwd_ws = self.wgrad_ws = self.stat_wgs = self.dgrad_bn = None
        self.pair_ws = {}  # images of the second segment -> workspace bytes of the paired weight gradient


def _desc(N, H, W, C1, C2, Cout, mode, prologue, dt, ld1, ld2, ldo, split_c=0, ldo2=0) -> _Plan:
    key = (N, H, W, C1, C2, Cout, mode, prologue, dt, ld1, ld2, ldo, split_c, ldo2)
    p = _desc_cache.get(key)
    if p is None:
        d = ConvDesc()
        d.N, d.H, d.W, d.C1, d.C2, d.Cout = N, H, W, C1, C2, Cout
        d.mode1, d.prologue, d.in_dtype, d.out_dtype = mode, prologue, dt, dt
        d.ld1, d.ld2, d.ldo, d.split_c, d.ldo2 = ld1, ld2, ldo, split_c, ldo2
        p = _desc_cache[key] = _Plan(d)
    return p


KERNEL_NAMES = {0: "conv3x3_igemm_kernel", 1: "conv3x3_plane_kernel", 4: "conv3x3_stream_kernel", 5: "conv3x3_flow_kernel"}


def conv3x3_plan(N, H, W, C1, C2, Cout, dtype: torch.dtype, mode: int = 0, prologue: bool = False) -> dict:
    """the launch plan of cy_conv3x3_fwd for this layer geometry (host-side query, no GPU needed)"""
    d = _desc(N, H, W, C1, C2, Cout, mode, 1 if prologue else 0, dtype_code(dtype), C1, C2, Cout)
    p = _lib.ConvPlan()
    _lib.call("cy_conv3x3_plan", d.ref, C.byref(p))
    out = {f: getattr(p, f) for f, _ in p._fields_}
    out["kernel"] = KERNEL_NAMES[out["kernel"]]
    return out


def conv3x3_wgrad_plan(N, H, W, C1, C2, Cout, dtype: torch.dtype, mode: int = 0, prologue: bool = False,
                       n_b: int = 0) -> dict:
    """the launch plan of cy_conv3x3_wgrad (n_b > 0: of cy_conv3x3_wgrad_pair) for this layer geometry"""
    d = _desc(N, H, W, C1, C2, Cout, mode, 1 if prologue else 0, dtype_code(dtype), C1, C2, Cout)
    p = _lib.WgradPlan()
    _lib.call("cy_conv3x3_wgrad_plan", d.ref, n_b, C.byref(p))
    return {f: getattr(p, f) for f, _ in p._fields_}


def _out_hw(src1: Tensor, mode: int) -> Tuple[int, int]:
    H, W = src1.shape[2], src1.shape[3]
    if mode == CY_SRC_POOL2:
        if H % 2 or W % 2:
            raise ValueError(f"max-pool 2x2 needs even spatial dims, got {H}x{W}")
        return H // 2, W // 2
    if mode == CY_SRC_UP2:
        return H * 2, W * 2
    return H, W


def conv3x3_fwd(src1: Tensor, src2: Optional[Tensor], wf: Tensor, Cout: int, *, mode: int = 0,
                scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
                want_stats: bool = True, split: Optional[int] = None, fold: Optional["BnState"] = None,
                stats_acc: bool = False):
    """out = conv3x3(cat(src1', src2)).  Returns (out, partials|None) or, with `split`,
    ((out[:, :split], out[:, split:]), None) as two separate NHWC tensors.
    `fold`: the BN+ReLU coefficients of source 1 come from the previous layer's accumulator (instead of scale / shift).
    `stats_acc`: the statistics go into a new accumulator: the second return value is (acc, R) instead of partial rows."""
    require_gpu(src1, wf)
    N, C1 = src1.shape[0], src1.shape[1]
    C2 = 0 if src2 is None else src2.shape[1]
    H, W = _out_hw(src1, mode)
    dt = dtype_code(src1.dtype)
    dev = src1.device
    prologue = 1 if (scale is not None or fold is not None) else 0
    if split:
        out = empty_nhwc(N, split, H, W, src1.dtype, dev)
        out2 = empty_nhwc(N, Cout - split, H, W, src1.dtype, dev)
        d = _desc(N, H, W, C1, C2, Cout, mode, prologue, dt, C1, C2, split, split, Cout - split)
    else:
        out = empty_nhwc(N, Cout, H, W, src1.dtype, dev)
        out2 = None
        d = _desc(N, H, W, C1, C2, Cout, mode, prologue, dt, C1, C2, Cout)
    stats = acc = None
    if want_stats and stats_acc:
        if d.stat_wgs is None:
            d.stat_wgs = _lib.call("cy_conv3x3_stat_workgroups", d.ref)
        acc = bn_acc_new(Cout, d.stat_wgs, dev)
    elif want_stats:
        if d.npart is None:
            d.npart = _lib.call("cy_conv3x3_num_partials", d.ref)
        stats = _f32(d.npart * 2 * Cout, dev).view(d.npart, 2, Cout)
    if d.fwd_ws is None:
        d.fwd_ws = _lib.load().cy_conv3x3_fwd_ws_bytes(d.ref)
    nbytes = d.fwd_ws
    ws = _ws(nbytes, dev) if nbytes else None
    ev = _prof_begin()
    if fold is not None or acc is not None:
        _lib.call("cy_conv3x3_fwd_bn", d.ref, src1.data_ptr(), _ptr(src2), None if fold is None else fold.ref,
                  _ptr(scale), _ptr(shift), wf.data_ptr(), out.data_ptr(), _ptr(out2), None,
                  None if acc is None else acc.ref, _ptr(ws), nbytes, _stream())
        stats = acc
    else:
        _lib.call("cy_conv3x3_fwd", d.ref, src1.data_ptr(), _ptr(src2), _ptr(scale), _ptr(shift),
                  wf.data_ptr(), out.data_ptr(), _ptr(out2), _ptr(stats), _ptr(ws), nbytes, _stream())
    if ev is not None:  # algorithmic bytes: every input and output element once, packed weights once
        esz = src1.element_size()
        nb = esz * (src1.numel() + (0 if src2 is None else src2.numel()) + N * H * W * Cout + 9 * (C1 + C2) * Cout)
        _prof_end(ev, "conv3x3_fwd_dgrad", 2.0 * N * H * W * 9 * (C1 + C2) * Cout, float(nb))
    if split:
        return (out, out2), None
    return out, stats


# ---- side stream for weight gradients ------------------------------------------------------------
# A layer's weight-gradient kernels do not feed the rest of the backward pass (only the optimizer
# reads them), and at the configured batch sizes neither they nor the data-gradient kernels fill
# 256 CUs on their own.  They are therefore enqueued on a second HIP stream and overlap with the
# main stream's data-gradient / BN-backward kernels; the optimizer joins the stream before it reads
# the gradients.  CY_ASYNC_WGRAD=0 keeps everything on one stream.

ASYNC_WGRAD = os.environ.get("CY_ASYNC_WGRAD", "1") != "0"
CAPTURING = False  # True while a HIP graph of the step is being captured (cyhip.graphed)
_side_streams = {}
_side_pending = set()


def side_stream(device, role: str = "wgrad") -> "torch.cuda.Stream":
    """the per-device auxiliary stream of a role: "wgrad" (weight gradients) or "pass2" (second
    network pass of a two-stage step)"""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    st = _side_streams.get((idx, role))
    if st is None:
        # "comm" waits on a word in memory (stream_wait_marks): a waiting packet stalls every stream that shares its
        # hardware queue, and streams of one priority are multiplexed onto a few queues -- a stream of another
        # priority cannot share one with the streams that compute
        st = _side_streams[(idx, role)] = torch.cuda.Stream(device=idx, priority=-1 if role == "comm" else 0)
    return st


def note_side_work(stream) -> None:
    """register work on an auxiliary stream that the end-of-backward / optimizer join must wait for
    (call BEFORE enqueueing it: from here on in-place updates of shared buffers are event-ordered)"""
    global _multi_stream_live
    _multi_stream_live = True
    _side_pending.add(stream)


class on_side_stream:
    """run the enclosed launches on the side stream, after everything already enqueued on the
    current stream; `reads` are tensors the side kernels read (kept alive for the allocator)"""

    def __init__(self, *reads: Optional[Tensor]):
        self.reads = [t for t in reads if t is not None]

    def __enter__(self):
        dev = self.reads[0].device
        self.side = side_stream(dev)
        self.side.wait_stream(torch.cuda.current_stream(dev))
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self.side

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        if not CAPTURING:  # inside a graph capture lifetimes are the graph's; record_stream is not capturable
            for t in self.reads:
                t.record_stream(self.side)
        _side_pending.add(self.side)
        global _join_queued
        if not _join_queued:  # join when this backward pass ends: .grad is then safe to read on the main stream
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_join_after_backward)
                _join_queued = True
            except RuntimeError:  # not inside a backward pass
                join_side_streams()
        return False


_join_queued = False


def ensure_backward_join() -> None:
    """called by backward nodes: if auxiliary streams carry work of this step (second-pass stream,
    weight-gradient stream) or work has been deferred to the end of the backward pass, make sure the
    deferred work is issued and the streams are joined when the backward pass ends"""
    global _join_queued
    if (_side_pending or _at_backward_end) and not _join_queued:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_after_backward)
            _join_queued = True
        except RuntimeError:
            pass


_at_backward_end = []  # callables to run once when the current backward pass ends (before the joins)


def at_backward_end(fn) -> None:
    """run `fn()` when the backward pass that is executing ends (immediately if none is)"""
    global _join_queued
    if fn not in _at_backward_end:
        _at_backward_end.append(fn)
    if not _join_queued:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_after_backward)
            _join_queued = True
        except RuntimeError:  # not inside a backward pass
            _join_after_backward()


def _join_after_backward() -> None:
    global _join_queued
    _join_queued = False
    while _at_backward_end:
        _at_backward_end.pop(0)()
    join_side_streams()


_home_stream = {}  # device index -> the stream the step itself runs on


def note_home_stream(device) -> None:
    """remember the stream the network is driven from (called by the forward pass); the joins below
    target it explicitly because autograd's end-of-backward callbacks may run on a thread whose
    current stream is the legacy default stream"""
    cur = torch.cuda.current_stream(device)
    if cur not in _side_streams.values():
        _home_stream[cur.device.index] = cur


def join_side_streams(onto: Optional["torch.cuda.Stream"] = None) -> None:
    """make `onto` (default: the home stream of the device) wait for all side-stream work enqueued so far"""
    while _side_pending:
        st = _side_pending.pop()
        tgt = onto if onto is not None else _home_stream.get(st.device.index)
        if tgt is None:
            tgt = torch.cuda.current_stream(st.device)
        tgt.wait_stream(st)


# ---- "these gradients are final" marks for the data-parallel optimizer ----------------------------------------------
# The backward pass finishes the decoder's and the deep encoder blocks' gradients (85 % of the parameters) long before
# it ends.  A block whose backward has run for the LAST time in a step leaves a mark; FusedRAdam starts the all-reduce
# of the buckets whose parameters are all marked on a communication stream that waits for those marks only, while the
# rest of the backward pass -- eager, or a replayed HIP graph -- still runs.
# A mark must be visible from OUTSIDE a captured graph.  (CUDA's answer, an external event-record node, is refused by
# this runtime: hipEventRecordWithFlags(hipEventRecordExternal) returns hipErrorInvalidValue under capture.)  So a mark
# is a word in device memory: every step has a serial number, kept in a device scalar the host refreshes before the
# step's first pass; the mark is a tiny copy "flag[tag] <- serial" on a helper stream that waits for every stream
# carrying work of the step (an ordinary kernel + fork / join, capturable); the consumer stream waits until
# flag[tag] >= serial (hipStreamWaitValue32).
# OFF unless CY_DP_EARLY=1: correct in the two-rank rehearsals (tests/test_gpu_distributed.py), but in one process on a
# one-rank RCCL group (tools/dp_single_rank.py) the early start costs 0.24 ms of step time -- about what it could
# hide at N = 8.  Not something to switch on without a multi-GPU measurement.
# CY_DP_EARLY: "0" / "1" force it; unset = the rule of `dp_early_rule` (applied by FusedRAdam when its flat buffers are
# built, i.e. before any step is captured).
DP_EARLY = os.environ.get("CY_DP_EARLY", "0") == "1"
DP_EARLY_FORCED = os.environ.get("CY_DP_EARLY") in ("0", "1")


def dp_early_rule(grad_bytes: int, world: int) -> bool:
    """Start gradient buckets inside the backward pass only where the all-reduce they would otherwise expose behind it
    is worth more than the early start costs.  Measured cost of the early start on a one-rank RCCL group: 0.24 ms per
    step (tools/dp_single_rank.py).  Exposed time of the late order, estimated: a ring all-reduce
    moves 2 (w - 1) / w of the bytes per GPU; RCCL's multi-ring schedule over the seven xGMI links of a node reaches
    ~400 GB/s per GPU for tens of MB, plus ~2 x 30 us of collective latency.  On iff that estimate exceeds 0.3 ms:
    the 34.5 MB of U-Net gradients at 8 ranks give 0.21 ms -> off; models from ~55 MB of gradients up switch it on."""
    if world <= 1:
        return False
    exposed = 2.0 * (world - 1) / world * grad_bytes / 400e9 + 60e-6
    return exposed > 0.3e-3
marks_wanted = False   # set by a data-parallel FusedRAdam
MARK_TAGS = ("decoder", "conv5", "conv4")
_step_id = {}          # device index -> [int32 device scalar, host value]
_mark_flags = {}       # (device index, tag) -> int32 device scalar
_ready_marks = {}      # tag -> flag tensor, for the marks left in the current step


def _dev_index(device) -> int:
    idx = torch.device(device).index
    return torch.cuda.current_device() if idx is None else idx


def begin_step_marks(device) -> None:
    """a new step (forward + backward) begins on `device`: bump its serial.  Outside graph capture only."""
    if not (marks_wanted and DP_EARLY) or CAPTURING:
        return
    idx = _dev_index(device)
    rec = _step_id.get(idx)
    if rec is None:
        rec = _step_id[idx] = [torch.zeros(1, dtype=torch.int32, device=f"cuda:{idx}"), 0]
        for tag in MARK_TAGS:
            _mark_flags[(idx, tag)] = torch.zeros(1, dtype=torch.int32, device=f"cuda:{idx}")
    rec[1] += 1
    rec[0].fill_(rec[1])
    _ready_marks.clear()


_MARK_AT = MARK_TAGS


def grad_ready_mark(tag: str, device) -> None:
    if not (marks_wanted and DP_EARLY) or tag not in _MARK_AT:
        return
    idx = _dev_index(device)
    rec = _step_id.get(idx)
    if rec is None:
        return
    flag = _mark_flags[(idx, tag)]
    cur = torch.cuda.current_stream(idx)
    if torch.cuda.is_current_stream_capturing():
        # Inside the captured step the weight gradients stay on their pass's stream, and this stream has already
        # waited for whatever the other pass contributes to the block's parameters (the parked operands' events of the
        # paired weight gradients, recorded after that pass's BatchNorm finalize; the `ordered` accumulations): the
        # mark is one 4-byte copy in this stream's own order.  (Forked onto a helper stream it cost 0.35 ms of graph
        # time per mark -- cross-stream edges are expensive in a replayed graph.)
        flag.copy_(rec[0])
        _ready_marks[tag] = flag
        return
    helper = side_stream(idx, "mark")
    streams = {cur, _home_stream.get(idx, cur)} | {st for (d, role), st in _side_streams.items()
                                                   if d == idx and role not in ("mark", "comm")}
    for st in streams:  # (eager mode only: the capturing case returned above)
        ev = torch.cuda.Event()
        ev.record(st)
        helper.wait_event(ev)
    with torch.cuda.stream(helper):
        flag.copy_(rec[0])
    note_side_work(helper)  # joined when the backward pass ends (inside the capture, if there is one)
    ensure_backward_join()
    _ready_marks[tag] = flag


def take_ready_marks() -> dict:
    """the marks of the step (tag -> flag), handing them over: the next step starts without any"""
    m = dict(_ready_marks)
    _ready_marks.clear()
    return m


def set_ready_marks(marks: dict) -> None:
    """a replayed backward graph leaves the marks of its capture"""
    _ready_marks.clear()
    _ready_marks.update(marks)


def clear_ready_marks() -> None:
    _ready_marks.clear()


def stream_wait_marks(stream, flags, device) -> None:
    """`stream` waits until every flag carries the serial of the current step"""
    serial = _step_id[_dev_index(device)][1]
    for flag in flags:
        _lib.call("cy_stream_wait_value", stream.cuda_stream, flag.data_ptr(), serial)


# ---- cross-stream ordering of read-modify-write kernels -----------------------------------------
# With the two network passes of a step on two streams (SemiSupervisedEpocher two-stage forward) the
# kernels that update a shared buffer in place -- BN running statistics, accumulated BN parameter
# gradients, batch counters -- must keep the reference's order: every such launch waits for the last
# launch on the same buffer (if that was on another stream) and leaves an event behind.  The order is
# the host's enqueue order, so results stay deterministic.
TWO_STREAM = os.environ.get("CY_TWO_STREAM", "1") != "0"
_order_events = {}
_multi_stream_live = False  # set by the first fork onto a "pass2" stream; single-stream runs pay nothing


class ordered:
    """`with ordered(key):` -- serialise in-place updates of the buffer identified by `key` across streams"""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        if _multi_stream_live:
            rec = _order_events.get(self.key)
            if rec is not None:
                cur = torch.cuda.current_stream()
                # an event recorded outside a graph capture cannot be waited on inside one, nor one of
                # another capture; captures are fenced by full stream synchronisation, so skipping is safe
                if rec[1] != cur and rec[2] == _capture_id(cur):
                    cur.wait_event(rec[0])
        return self

    def __exit__(self, *exc):
        if _multi_stream_live:
            rec = _order_events.get(self.key)
            cur = torch.cuda.current_stream()
            cap = _capture_id(cur)
            ev = rec[0] if (rec is not None and rec[2] == cap) else torch.cuda.Event()
            ev.record(cur)
            _order_events[self.key] = (ev, cur, cap)
        return False


def _capture_id(stream) -> int:
    """0 outside HIP-graph capture, else the id of the capture `stream` belongs to"""
    if not torch.cuda.is_current_stream_capturing():
        return 0
    return int(_lib.load().cy_stream_capture_id(stream.cuda_stream))


def grad_sink(p: Tensor) -> Optional[Tensor]:
    """p.grad when the kernels can accumulate straight into it (f32, contiguous, on the GPU)"""
    if p is None or not p.is_leaf:
        return None
    g = p.grad
    if g is not None and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous():
        p.__dict__["_cy_touched"] = True  # read by FusedRAdam: this parameter received a gradient
        return g
    return None


def conv3x3_wgrad(src1: Tensor, src2: Optional[Tensor], dy: Tensor, *, mode: int = 0,
                  scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
                  out: Optional[Tensor] = None) -> Tensor:
    """dw [Cout,Cin,3,3] f32; with `out` the result is ADDED into out (gradient accumulation)."""
    require_gpu(src1, dy)
    N, C1 = src1.shape[0], src1.shape[1]
    C2 = 0 if src2 is None else src2.shape[1]
    Cout, H, W = dy.shape[1], dy.shape[2], dy.shape[3]
    dt = dtype_code(src1.dtype)
    d = _desc(N, H, W, C1, C2, Cout, mode, 1 if scale is not None else 0, dt, C1, C2, Cout)
    if d.wgrad_ws is None:
        d.wgrad_ws = _lib.load().cy_conv3x3_wgrad_ws_bytes(d.ref)
    nbytes = d.wgrad_ws
    ws = _ws(nbytes, src1.device)
    dw = out if out is not None else torch.empty((Cout, C1 + C2, 3, 3), dtype=torch.float32, device=src1.device)
    ev = _prof_begin()
    # accumulation into a live .grad is a read-modify-write: the two passes of a step may reach the
    # same weight from different streams
    with (ordered(("conv_grad", out.data_ptr())) if out is not None else contextlib.nullcontext()):
        _lib.call("cy_conv3x3_wgrad", d.ref, src1.data_ptr(), _ptr(src2), _ptr(scale), _ptr(shift),
                  dy.data_ptr(), dw.data_ptr(), 0 if out is None else 1, ws.data_ptr(), nbytes, _stream())
    if ev is not None:
        esz = src1.element_size()
        nb = esz * (src1.numel() + (0 if src2 is None else src2.numel()) + dy.numel()) + 4 * 9 * (C1 + C2) * Cout
        _prof_end(ev, "conv3x3_wgrad", 2.0 * N * H * W * 9 * (C1 + C2) * Cout, float(nb))
    return dw


def conv3x3_wgrad_pair(src1: Tensor, src2: Optional[Tensor], dy: Tensor, scale: Optional[Tensor],
                       shift: Optional[Tensor], src1_b: Tensor, src2_b: Optional[Tensor], dy_b: Tensor,
                       scale_b: Optional[Tensor], shift_b: Optional[Tensor], *, mode: int, out: Tensor) -> Tensor:
    """out += dw(segment a) + dw(segment b): the same layer on two batches (e.g. the two passes of a
    two-stage step) in ONE launch.  bf16 tensors of equal geometry per image."""
    require_gpu(src1, dy, src1_b, dy_b)
    N, C1 = src1.shape[0], src1.shape[1]
    C2 = 0 if src2 is None else src2.shape[1]
    Cout, H, W = dy.shape[1], dy.shape[2], dy.shape[3]
    if tuple(src1_b.shape[1:]) != tuple(src1.shape[1:]) or tuple(dy_b.shape[1:]) != tuple(dy.shape[1:]) \
            or src1_b.dtype != src1.dtype or (src2 is None) != (src2_b is None) or (scale is None) != (scale_b is None):
        raise ValueError("conv3x3_wgrad_pair: the two segments must be the same layer")
    d = _desc(N, H, W, C1, C2, Cout, mode, 1 if scale is not None else 0, dtype_code(src1.dtype), C1, C2, Cout)
    nb = src1_b.shape[0]
    nbytes = d.pair_ws.get(nb)
    if nbytes is None:
        nbytes = d.pair_ws[nb] = _lib.load().cy_conv3x3_wgrad_pair_ws_bytes(d.ref, nb)
    ws = _ws(nbytes, src1.device)
    ev = _prof_begin()
    with ordered(("conv_grad", out.data_ptr())):
        _lib.call("cy_conv3x3_wgrad_pair", d.ref, src1.data_ptr(), _ptr(src2), _ptr(scale), _ptr(shift),
                  dy.data_ptr(), nb, src1_b.data_ptr(), _ptr(src2_b), _ptr(scale_b), _ptr(shift_b),
                  dy_b.data_ptr(), out.data_ptr(), 1, ws.data_ptr(), nbytes, _stream())
    if ev is not None:
        esz = src1.element_size()
        nbts = esz * sum(t.numel() for t in (src1, src2, dy, src1_b, src2_b, dy_b) if t is not None) \
            + 4 * 9 * (C1 + C2) * Cout
        _prof_end(ev, "conv3x3_wgrad", 2.0 * (N + nb) * H * W * 9 * (C1 + C2) * Cout, float(nbts))
    return out


def conv_first_fwd(x: Tensor, w: Tensor, out_dtype: torch.dtype, want_stats: bool = True, stats_acc: bool = False):
    """x: f32 NCHW image [N,Cin<=4,H,W]; w: f32 [Cout,Cin,3,3]."""
    require_gpu(x, w)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    x = x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()
    w = w.detach().contiguous()
    out = empty_nhwc(N, Cout, H, W, out_dtype, x.device)
    stats = None
    if want_stats and stats_acc:
        npart = _lib.call("cy_conv3x3_first_num_partials", N, H, W, Cout)
        acc = bn_acc_new(Cout, npart, x.device)
        _lib.call("cy_conv3x3_first_fwd_acc", x.data_ptr(), w.data_ptr(), out.data_ptr(), acc.ref, N, Cin,
                  H, W, Cout, dtype_code(out_dtype), _stream())
        return out, acc
    if want_stats:
        npart = _lib.call("cy_conv3x3_first_num_partials", N, H, W, Cout)
        stats = _f32(npart * 2 * Cout, x.device).view(npart, 2, Cout)
    _lib.call("cy_conv3x3_first_fwd", x.data_ptr(), w.data_ptr(), out.data_ptr(), _ptr(stats), N, Cin,
              H, W, Cout, dtype_code(out_dtype), _stream())
    return out, stats


def conv_first_wgrad(x: Tensor, dy: Tensor, out: Optional[Tensor] = None) -> Tensor:
    require_gpu(x, dy)
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    x = x.contiguous() if x.dtype == torch.float32 else x.float().contiguous()
    nbytes = _lib.load().cy_conv3x3_first_wgrad_ws_bytes(N, Cin, H, W, Cout)
    ws = _ws(nbytes, x.device)
    dw = out if out is not None else torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device)
    with (ordered(("conv_grad", out.data_ptr())) if out is not None else contextlib.nullcontext()):
        _lib.call("cy_conv3x3_first_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0 if out is None else 1,
                  N, Cin, H, W, Cout, dtype_code(dy.dtype), ws.data_ptr(), nbytes, _stream())
    return dw


# --------------------------------------------------------------------------- BN + ReLU
# BatchNorm sums without finalize launches (csrc/cy_bn_acc.h): producers add into a zeroed int64 accumulator, consumers
# derive the coefficients themselves.  CY_BN_ACC=0 keeps the partial rows + finalize launches of rounds 1-3 (A/B runs).
BN_ACC = os.environ.get("CY_BN_ACC", "1") != "0"


class BnArena:
    """zeroed int64 words for the accumulators of one network pass: ONE fill launch per pass instead of one per layer.
    The size follows what earlier passes used; an allocation that does not fit gets its own zeroed tensor."""
    high_water = 0

    def __init__(self, device):
        self.used = 0
        n = BnArena.high_water
        self.buf = torch.zeros(n, dtype=torch.int64, device=device) if n else None

    def alloc(self, words: int, device) -> Tensor:
        a = self.used
        self.used += (words + 15) & ~15  # (128-byte granules: no line is shared by two accumulators)
        if self.used > BnArena.high_water:
            BnArena.high_water = self.used
        if self.buf is not None and self.used <= self.buf.numel() and self.buf.device == torch.device(device):
            return self.buf[a:a + words]
        return torch.zeros(words, dtype=torch.int64, device=device)


_bn_arena: Optional[BnArena] = None


def bn_arena_begin(device) -> Optional[BnArena]:
    """called when a network pass begins; returns the previous arena (restored by `bn_arena_end`)"""
    global _bn_arena
    prev, _bn_arena = _bn_arena, (BnArena(device) if BN_ACC else None)
    return prev


def bn_arena_end(prev: Optional[BnArena]) -> None:
    global _bn_arena
    _bn_arena = prev


class BnAccBuf:
    """an accumulator [R][4][C] (+ R flags) and its ctypes view"""
    __slots__ = ("t", "R", "C", "s", "ref", "filled_for")

    def __init__(self, t: Tensor, R: int, Cc: int):
        self.t, self.R, self.C = t, R, Cc
        self.filled_for = None  # the gradient tensor whose producer has added its sums already
        self.s = _lib.BnAcc(t.data_ptr(), R, Cc)
        self.ref = C.byref(self.s)


def bn_acc_new(Cc: int, workgroups: int, device) -> BnAccBuf:
    R = _lib.call("cy_bn_acc_replicas", Cc, max(1, workgroups))
    words = R * 4 * Cc + R
    t = _bn_arena.alloc(words, device) if _bn_arena is not None else torch.zeros(words, dtype=torch.int64, device=device)
    return BnAccBuf(t, R, Cc)


class BnState:
    """one training-mode BatchNorm evaluation on an accumulator: what its consumer needs to derive relu(scale*y+shift)
    (cy_bn_fold) and where it leaves [scale, shift, mean, invstd, unbiased var] for the backward pass"""
    __slots__ = ("acc", "gamma", "beta", "count", "eps", "coef", "s", "ref", "done")

    def __init__(self, acc: BnAccBuf, gamma: Optional[Tensor], beta: Optional[Tensor], count: int, eps: float, device):
        self.acc, self.gamma, self.beta, self.count, self.eps = acc, gamma, beta, count, eps
        self.coef = _f32(5 * acc.C, device).view(5, acc.C)
        self.s = _lib.BnFold(acc.t.data_ptr(), acc.R, acc.C, _ptr(gamma), _ptr(beta), float(count), float(eps), 0,
                             self.coef.data_ptr())
        self.ref = C.byref(self.s)
        self.done = False  # a consumer has written `coef`


def bn_fold_coef(st: BnState) -> None:
    """accumulator -> st.coef as its own launch (when no folding consumer follows)"""
    _lib.call("cy_bn_fold_coef", st.ref, _stream())
    st.done = True


def bn_relu_apply_fold(y: Tensor, st: BnState, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    N, Cc, H, W = y.shape
    out_dtype = out_dtype or y.dtype
    out = empty_nhwc(N, Cc, H, W, out_dtype, y.device)
    _lib.call("cy_bn_relu_apply_fold", y.data_ptr(), st.ref, out.data_ptr(), N * H * W, dtype_code(y.dtype),
              dtype_code(out_dtype), _stream())
    st.done = True
    return out


def bn_relu_apply_pool_fold(y: Tensor, st: BnState) -> Tuple[Tensor, Tensor]:
    N, Cc, H2, W2 = y.shape
    if H2 % 2 or W2 % 2:
        raise ValueError(f"max-pool 2x2 needs even spatial dims, got {H2}x{W2}")
    out = empty_nhwc(N, Cc, H2, W2, y.dtype, y.device)
    pooled = empty_nhwc(N, Cc, H2 // 2, W2 // 2, y.dtype, y.device)
    dt = dtype_code(y.dtype)
    _lib.call("cy_bn_relu_apply_pool_fold", y.data_ptr(), st.ref, out.data_ptr(), pooled.data_ptr(), N, H2 // 2,
              W2 // 2, dt, dt, _stream())
    st.done = True
    return out, pooled


def bn_running_update(items) -> None:
    """items: [(coef [5, C], running_mean, running_var, momentum)] -- the running statistics of all of them in one
    launch per 32 layers, event-ordered against the other pass of a two-stream step like the per-layer updates were"""
    if not items:
        return
    arr = (_lib.BnRunItem * len(items))()
    for a, (coef, rm, rv, mom) in zip(arr, items):
        a.coef, a.running_mean, a.running_var, a.C, a.momentum = coef.data_ptr(), rm.data_ptr(), rv.data_ptr(), coef.shape[1], mom
    with ordered("bn_running_batch"):
        _lib.call("cy_bn_running_update", arr, len(items), _stream())


def bn_finalize(partials: Optional[Tensor], count: int, gamma: Optional[Tensor], beta: Optional[Tensor],
                running_mean: Optional[Tensor], running_var: Optional[Tensor], momentum: float,
                eps: float, use_batch_stats: bool, update_running: bool, Cc: int, device):
    out = _f32(4 * Cc, device).view(4, Cc)
    npart = 0 if partials is None else partials.shape[0]
    def launch():
        _lib.call("cy_bn_finalize", _ptr(partials), npart, Cc, float(count), _ptr(gamma), _ptr(beta),
                  _ptr(running_mean), _ptr(running_var), float(momentum), float(eps),
                  int(use_batch_stats), int(update_running), out[0].data_ptr(), out[1].data_ptr(),
                  out[2].data_ptr(), out[3].data_ptr(), _stream())

    if running_mean is not None and (update_running or not use_batch_stats):
        # reads or updates the running statistics (the batched updates of the accumulator path share the second key)
        with ordered(("bn_running", running_mean.data_ptr())), ordered("bn_running_batch"):
            launch()
    else:
        launch()
    return out[0], out[1], out[2], out[3]  # scale, shift, mean, invstd


def bn_relu_apply(y: Tensor, scale: Tensor, shift: Tensor, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    N, Cc, H, W = y.shape
    out_dtype = out_dtype or y.dtype
    out = empty_nhwc(N, Cc, H, W, out_dtype, y.device)
    _lib.call("cy_bn_relu_apply", y.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
              N * H * W, Cc, dtype_code(y.dtype), dtype_code(out_dtype), _stream())
    return out


def bn_relu_apply_pool(y: Tensor, scale: Tensor, shift: Tensor) -> Tuple[Tensor, Tensor]:
    """(relu(scale*y + shift), its 2x2 max) in one pass -- the block output and what nn.MaxPool2d(2) makes of it"""
    N, Cc, H2, W2 = y.shape
    if H2 % 2 or W2 % 2:
        raise ValueError(f"max-pool 2x2 needs even spatial dims, got {H2}x{W2}")
    out = empty_nhwc(N, Cc, H2, W2, y.dtype, y.device)
    pooled = empty_nhwc(N, Cc, H2 // 2, W2 // 2, y.dtype, y.device)
    dt = dtype_code(y.dtype)
    _lib.call("cy_bn_relu_apply_pool", y.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
              pooled.data_ptr(), N, H2 // 2, W2 // 2, Cc, dt, dt, _stream())
    return out, pooled


def bn_relu_bwd(da: Tensor, y: Tensor, scale: Tensor, shift: Tensor, mean: Tensor, invstd: Tensor,
                batch_stats: bool, dgamma_out: Optional[Tensor] = None, dbeta_out: Optional[Tensor] = None,
                want_param_grads: bool = True, partials: Optional[Tensor] = None):
    """Backward of a = relu(bn(y)): returns (dy, dgamma, dbeta).  With dgamma_out/dbeta_out the
    parameter gradients are ADDED into those buffers (and returned as None).  `partials`: the reduction pass
    was already done by the kernel that produced `da` (maxpool2_bwd_bn)."""
    N, Cc, H, W = y.shape
    npix = N * H * W
    dev = y.device
    dt = dtype_code(y.dtype)
    if da.dtype != y.dtype:
        da = da.to(y.dtype)
    da = to_nhwc(da)
    if partials is not None:
        part, npart = partials, partials.shape[0]
    else:
        npart = _lib.call("cy_bn_bwd_num_partials", npix, Cc)
        part = _f32(npart * 2 * Cc, dev)
        _lib.call("cy_bn_relu_bwd_reduce", da.data_ptr(), Cc, y.data_ptr(), scale.data_ptr(),
                  shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), part.data_ptr(), npix, Cc, dt,
                  _stream())
    coef = _f32(2 * Cc, dev)
    acc = dgamma_out is not None
    dgamma = dbeta = None
    if acc:
        pg, pb = dgamma_out, dbeta_out
    elif want_param_grads:
        gb = _f32(2 * Cc, dev).view(2, Cc)
        dgamma, dbeta = gb[0], gb[1]
        pg, pb = dgamma, dbeta
    else:
        pg = pb = None
    def finalize():
        _lib.call("cy_bn_bwd_finalize", part.data_ptr(), npart, Cc, scale.data_ptr(), mean.data_ptr(),
                  invstd.data_ptr(), float(npix), int(batch_stats), _ptr(pg), _ptr(pb), int(acc), coef.data_ptr(),
                  _stream())

    if acc:
        with ordered(("bn_grad", pg.data_ptr())):  # in-place accumulation into the parameters' .grad
            finalize()
    else:
        finalize()
    dy = empty_nhwc(N, Cc, H, W, y.dtype, dev)
    _lib.call("cy_bn_relu_bwd_apply", da.data_ptr(), Cc, y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
              coef.data_ptr(), dy.data_ptr(), npix, Cc, dt, _stream())
    return dy, dgamma, dbeta


POOL_BN_FUSE = os.environ.get("CY_POOL_BN_FUSE", "1") != "0"  # (A/B switch)


def bn_relu_bwd_acc(da: Tensor, y: Tensor, coef: Tensor, batch_stats: bool, dgamma_out: Optional[Tensor] = None,
                    dbeta_out: Optional[Tensor] = None, want_param_grads: bool = True,
                    acc: Optional[BnAccBuf] = None, acc_filled: bool = False):
    """bn_relu_bwd on an accumulator: the reduce launch adds the sums of dz and dz*xhat into `acc` (skipped when the
    kernel that produced `da` already did, `acc_filled`), the apply launch derives (k1, k0) from it and its first
    workgroup adds the parameter gradients -- two launches instead of three.  coef: the forward pass's [5, C]."""
    N, Cc, H, W = y.shape
    npix = N * H * W
    dev = y.device
    dt = dtype_code(y.dtype)
    if da.dtype != y.dtype:
        da = da.to(y.dtype)
    da = to_nhwc(da)
    if acc is None:
        acc = bn_acc_new(Cc, _lib.call("cy_bn_relu_bwd_workgroups", npix, Cc), dev)
    if not acc_filled:
        _lib.call("cy_bn_relu_bwd_reduce_acc", da.data_ptr(), Cc, y.data_ptr(), coef.data_ptr(), acc.ref, npix, Cc, dt,
                  _stream())
    accum = dgamma_out is not None
    dgamma = dbeta = None
    if accum:
        pg, pb = dgamma_out, dbeta_out
    elif want_param_grads:
        gb = _f32(2 * Cc, dev).view(2, Cc)
        dgamma, dbeta = gb[0], gb[1]
        pg, pb = dgamma, dbeta
    else:
        pg = pb = None
    dy = empty_nhwc(N, Cc, H, W, y.dtype, dev)

    def apply():
        _lib.call("cy_bn_relu_bwd_apply_fold", da.data_ptr(), Cc, y.data_ptr(), coef.data_ptr(), acc.ref, float(npix),
                  int(batch_stats), _ptr(pg), _ptr(pb), int(accum), dy.data_ptr(), npix, Cc, dt, _stream())

    if accum:
        with ordered(("bn_grad", pg.data_ptr())):  # in-place accumulation into the parameters' .grad
            apply()
    else:
        apply()
    return dy, dgamma, dbeta


def bn_bwd_reduce_acc(da: Tensor, y: Tensor, coef: Tensor, acc: BnAccBuf) -> None:
    """acc += the sums of dz and dz*xhat over (da, y) (the reduce launch of bn_relu_bwd_acc on its own)"""
    N, Cc, H, W = y.shape
    _lib.call("cy_bn_relu_bwd_reduce_acc", da.data_ptr(), Cc, y.data_ptr(), coef.data_ptr(), acc.ref, N * H * W, Cc,
              dtype_code(y.dtype), _stream())


def bn_bwd_acc_new(N: int, Cc: int, H: int, W: int, pooled: bool, device) -> BnAccBuf:
    """a zeroed accumulator for the backward sums of a BatchNorm over [N, C, H, W], sized for whichever kernel will
    fill it (the reduce launch, the pool backward of the block's output, or the upsample backward of its consumer)"""
    wgs = _lib.call("cy_bn_relu_bwd_workgroups", N * H * W, Cc)
    if pooled and POOL_BN_FUSE and H % 2 == 0 and W % 2 == 0:
        wgs = max(wgs, _lib.load().cy_maxpool2_bwd_bn_num_partials(N, H // 2, W // 2, Cc))
    wgs = max(wgs, _lib.load().cy_upsample2_bwd_bn_workgroups(N, H, W, Cc))
    return bn_acc_new(Cc, wgs, device)


def upsample2_bwd_bn_acc(dup: Tensor, y: Tensor, coef: Tensor, acc: BnAccBuf) -> Optional[Tensor]:
    """upsample2_bwd whose result is the dA of relu(bn(y)): that BatchNorm's backward sums are added into `acc`;
    None where the fused form does not apply"""
    N, Cc, H2, W2 = dup.shape
    if _lib.load().cy_upsample2_bwd_bn_workgroups(N, H2 // 2, W2 // 2, Cc) <= 0 or dup.dtype != y.dtype:
        return None
    dx = empty_nhwc(N, Cc, H2 // 2, W2 // 2, dup.dtype, dup.device)
    _lib.call("cy_upsample2_bwd_bn_acc", dup.data_ptr(), Cc, dx.data_ptr(), y.data_ptr(), coef.data_ptr(), acc.ref, N,
              H2 // 2, W2 // 2, Cc, dtype_code(dup.dtype), _stream())
    return dx


def maxpool2_bwd_bn_acc(x: Tensor, dpool: Tensor, add: Optional[Tensor], y: Tensor, coef: Tensor,
                        acc: Optional[BnAccBuf] = None):
    """maxpool2_bwd whose result is the dA of relu(bn(y)), with that BatchNorm's backward sums added into `acc` (a new
    accumulator if None): (dx, acc), or (dx, None) where the fused form does not apply"""
    N, Cc, H2, W2 = x.shape
    npart = _lib.load().cy_maxpool2_bwd_bn_num_partials(N, H2 // 2, W2 // 2, Cc) if POOL_BN_FUSE else 0
    if npart <= 0:
        return maxpool2_bwd(x, dpool, add), None
    dx = empty_nhwc(N, Cc, H2, W2, x.dtype, x.device)
    if acc is None:
        acc = bn_acc_new(Cc, npart, x.device)
    _lib.call("cy_maxpool2_bwd_bn_acc", x.data_ptr(), dpool.data_ptr(), _ptr(add), Cc, dx.data_ptr(), y.data_ptr(),
              coef.data_ptr(), acc.ref, N, H2 // 2, W2 // 2, Cc, dtype_code(x.dtype), _stream())
    return dx, acc


# The data gradient with the BatchNorm + ReLU backward in its load path (cy_conv3x3_dgrad_bn): one launch instead of
# cy_bn_relu_bwd_apply_fold + cy_conv3x3_fwd; dy comes back as a second output for the weight gradient.  OFF unless
# CY_DGRAD_BN=1: measured per layer (tools/bench_dgrad_bn.py) and in the step (tools/ab.sh CY_DGRAD_BN 0 1) it does not
# pay -- every cout block of a tile, and every tile for its halo, repeats the in-LDS pass over (dA, y) that the apply
# launch does once per element; on the 128-cout tilings the fused launch is 3-5 us shorter than the two it replaces, on
# the 64-cout tilings 5-13 us longer, and the step does not move (6.51 against 6.51 ms with the 128-cout layers fused,
# 6.38 against 6.25 with all of them).  DESIGN.md section 3 "Round 4".
DGRAD_BN = os.environ.get("CY_DGRAD_BN", "0") == "1"


def _dgrad_bn_desc(da: Tensor, Cin: int, split: Optional[int]):
    N, Cc, H, W = da.shape
    dt = dtype_code(da.dtype)
    if split:
        return _desc(N, H, W, Cc, 0, Cin, 0, 2, dt, Cc, 0, split, split, Cin - split)
    return _desc(N, H, W, Cc, 0, Cin, 0, 2, dt, Cc, 0, Cin)


def conv3x3_dgrad_bn_ok(da: Tensor, Cin: int, split: Optional[int] = None) -> bool:
    """does this layer geometry have a launch plan that takes the fused form?"""
    if not (DGRAD_BN and BN_ACC) or da.dtype not in HALF_TYPES or not is_nhwc(da):
        return False
    d = _dgrad_bn_desc(da, Cin, split)
    if d.dgrad_bn is None:
        d.dgrad_bn = bool(_lib.call("cy_conv3x3_dgrad_bn_ok", d.ref))
    return d.dgrad_bn


def conv3x3_dgrad_bn(da: Tensor, y: Tensor, coef: Tensor, acc: "BnAccBuf", batch_stats: bool, wd: Tensor, Cin: int, *,
                     dgamma_out: Optional[Tensor] = None, dbeta_out: Optional[Tensor] = None,
                     want_param_grads: bool = True, split: Optional[int] = None):
    """(dx or (dx1, dx2), dy, dgamma, dbeta): dy = the BatchNorm + ReLU backward of da (formed in the conv's load path
    from da, y and the sums in `acc`), dx = conv3x3(dy, wd).  coef: row 0 of the forward pass's [5, C] block."""
    N, Cc, H, W = y.shape
    dev = y.device
    d = _dgrad_bn_desc(da, Cin, split)
    if split:
        out = empty_nhwc(N, split, H, W, y.dtype, dev)
        out2 = empty_nhwc(N, Cin - split, H, W, y.dtype, dev)
    else:
        out, out2 = empty_nhwc(N, Cin, H, W, y.dtype, dev), None
    dy = empty_nhwc(N, Cc, H, W, y.dtype, dev)
    accum = dgamma_out is not None
    dgamma = dbeta = None
    if accum:
        pg, pb = dgamma_out, dbeta_out
    elif want_param_grads:
        gb = _f32(2 * Cc, dev).view(2, Cc)
        dgamma, dbeta = gb[0], gb[1]
        pg, pb = dgamma, dbeta
    else:
        pg = pb = None
    if d.fwd_ws is None:
        d.fwd_ws = _lib.load().cy_conv3x3_fwd_ws_bytes(d.ref)
    nbytes = d.fwd_ws
    ws = _ws(nbytes, dev) if nbytes else None
    bn = _lib.BnBwdIn(y.data_ptr(), coef.data_ptr(), C.pointer(acc.s), float(N * H * W), int(batch_stats), int(accum),
                      _ptr(pg), _ptr(pb), dy.data_ptr())
    ev = _prof_begin()

    def launch():
        _lib.call("cy_conv3x3_dgrad_bn", d.ref, da.data_ptr(), C.byref(bn), wd.data_ptr(), out.data_ptr(), _ptr(out2),
                  _ptr(ws), nbytes, _stream())

    if accum:
        with ordered(("bn_grad", pg.data_ptr())):  # the first workgroup adds into the parameters' .grad
            launch()
    else:
        launch()
    if ev is not None:
        esz = y.element_size()
        nb = esz * (3 * y.numel() + N * H * W * Cin + 9 * Cc * Cin)
        _prof_end(ev, "conv3x3_fwd_dgrad", 2.0 * N * H * W * 9 * Cc * Cin, float(nb))
    return ((out, out2) if split else out), dy, dgamma, dbeta


# A data gradient whose output is the dA of a BatchNorm + ReLU adds that layer's backward sums in its epilogue
# (cy_conv3x3_dgrad_dz): the reduce launch over (dA, y) goes.  OFF unless CY_DGRAD_DZ=1: on the 16-row tilings the fused
# launch takes what the two took (28.0 against 20.6 + 7.5 us at 56 x 56 x 128, tools/bench_dgrad_dz.py), on the tilings
# with four fragments per wave the sums' working set spills ~900 registers (3-4 x slower: excluded by the plan), and the
# step does not move (tools/ab.sh CY_DGRAD_DZ 0 1: 6.28 against 6.28 ms) -- the conv kernels are what the step waits for,
# work moved into them costs what it saved.  DESIGN.md section 3 "Round 4".
DGRAD_DZ = os.environ.get("CY_DGRAD_DZ", "0") == "1"
_dz_ok_cache = {}


def conv3x3_dgrad_dz_ok(dy: Tensor, Cin: int, split: Optional[int], c0: int, Cs: int) -> bool:
    if not (DGRAD_DZ and BN_ACC) or dy.dtype not in HALF_TYPES:
        return False
    N, Cc, H, W = dy.shape
    key = (N, Cc, H, W, Cin, split, c0, Cs, dy.dtype)
    ok = _dz_ok_cache.get(key)
    if ok is None:
        dt = dtype_code(dy.dtype)
        d = _desc(N, H, W, Cc, 0, Cin, 0, 0, dt, Cc, 0, split, split, Cin - split) if split else \
            _desc(N, H, W, Cc, 0, Cin, 0, 0, dt, Cc, 0, Cin)
        ok = _dz_ok_cache[key] = bool(_lib.call("cy_conv3x3_dgrad_dz_ok", d.ref, c0, Cs))
    return ok


def conv3x3_dgrad_dz(dy: Tensor, wd: Tensor, Cin: int, y: Tensor, coef: Tensor, acc: "BnAccBuf", *,
                     split: Optional[int] = None):
    """conv3x3_fwd(dy, wd) as a data gradient whose output -- all of it, or with `split` its second part -- is the dA
    of relu(bn(y)): that layer's backward sums are added into `acc` by the epilogue.  Returns out or (out1, out2)."""
    N, Cc, H, W = dy.shape
    dev = dy.device
    dt = dtype_code(dy.dtype)
    if split:
        out = empty_nhwc(N, split, H, W, dy.dtype, dev)
        out2 = empty_nhwc(N, Cin - split, H, W, dy.dtype, dev)
        d = _desc(N, H, W, Cc, 0, Cin, 0, 0, dt, Cc, 0, split, split, Cin - split)
    else:
        out, out2 = empty_nhwc(N, Cin, H, W, dy.dtype, dev), None
        d = _desc(N, H, W, Cc, 0, Cin, 0, 0, dt, Cc, 0, Cin)
    if d.fwd_ws is None:
        d.fwd_ws = _lib.load().cy_conv3x3_fwd_ws_bytes(d.ref)
    nbytes = d.fwd_ws
    ws = _ws(nbytes, dev) if nbytes else None
    dz = _lib.BnDzOut(y.data_ptr(), coef.data_ptr(), C.pointer(acc.s), split or 0, y.shape[1])
    ev = _prof_begin()
    _lib.call("cy_conv3x3_dgrad_dz", d.ref, dy.data_ptr(), wd.data_ptr(), out.data_ptr(), _ptr(out2), C.byref(dz),
              _ptr(ws), nbytes, _stream())
    if ev is not None:
        esz = dy.element_size()
        nb = esz * (dy.numel() + y.numel() + N * H * W * Cin + 9 * Cc * Cin)
        _prof_end(ev, "conv3x3_fwd_dgrad", 2.0 * N * H * W * 9 * Cc * Cin, float(nb))
    return (out, out2) if split else out


def maxpool2_bwd(x: Tensor, dpool: Tensor, add: Optional[Tensor] = None) -> Tensor:
    N, Cc, H2, W2 = x.shape
    dx = empty_nhwc(N, Cc, H2, W2, x.dtype, x.device)
    _lib.call("cy_maxpool2_bwd", x.data_ptr(), dpool.data_ptr(), _ptr(add), Cc, dx.data_ptr(), N,
              H2 // 2, W2 // 2, Cc, dtype_code(x.dtype), _stream())
    return dx


def maxpool2_bwd_bn(x: Tensor, dpool: Tensor, add: Optional[Tensor], y: Tensor, scale: Tensor, shift: Tensor,
                    mean: Tensor, invstd: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
    """maxpool2_bwd whose result is the dA of relu(bn(y)): (dx, partial rows [P, 2, C] of that BatchNorm's backward
    sums), or (dx, None) where the fused form does not apply (channel groups that do not divide a workgroup)"""
    N, Cc, H2, W2 = x.shape
    npart = _lib.load().cy_maxpool2_bwd_bn_num_partials(N, H2 // 2, W2 // 2, Cc) if POOL_BN_FUSE else 0
    if npart <= 0:
        return maxpool2_bwd(x, dpool, add), None
    dx = empty_nhwc(N, Cc, H2, W2, x.dtype, x.device)
    part = _f32(npart * 2 * Cc, x.device).view(npart, 2, Cc)
    _lib.call("cy_maxpool2_bwd_bn", x.data_ptr(), dpool.data_ptr(), _ptr(add), Cc, dx.data_ptr(), y.data_ptr(),
              scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), part.data_ptr(), N,
              H2 // 2, W2 // 2, Cc, dtype_code(x.dtype), _stream())
    return dx, part


def upsample2_bwd(dup: Tensor) -> Tensor:
    N, Cc, H2, W2 = dup.shape
    dx = empty_nhwc(N, Cc, H2 // 2, W2 // 2, dup.dtype, dup.device)
    _lib.call("cy_upsample2_bwd", dup.data_ptr(), Cc, dx.data_ptr(), N, H2 // 2, W2 // 2, Cc,
              dtype_code(dup.dtype), _stream())
    return dx


# --------------------------------------------------------------------------- head + losses
def head_fwd(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """x [N,C,H,W] NHWC -> f32 logits [N,K,H,W] NHWC."""
    require_gpu(x, w)
    N, Cc, H, W = x.shape
    K = w.shape[0]
    logits = empty_nhwc(N, K, H, W, torch.float32, x.device)
    w2 = w.detach().reshape(K, Cc).float().contiguous()
    _lib.call("cy_head1x1_fwd", x.data_ptr(), w2.data_ptr(), _ptr(None if b is None else b.detach()),
              logits.data_ptr(), N * H * W, Cc, K, dtype_code(x.dtype), _stream())
    return logits


def head_bwd(x: Tensor, w: Tensor, dlogits: Tensor, need_dx: bool, need_dw: bool,
             dw_into: Optional[Tensor] = None, db_into: Optional[Tensor] = None):
    """(dx, dw, db); with dw_into (and db_into for a head with bias) the parameter gradients are ADDED into those live
    .grad buffers by the reduction kernel and returned as None"""
    N, Cc, H, W = x.shape
    K = w.shape[0]
    npix = N * H * W
    dlogits = to_nhwc(dlogits.float())
    w2 = w.detach().reshape(K, Cc).float().contiguous()
    dx = empty_nhwc(N, Cc, H, W, x.dtype, x.device) if need_dx else None
    dw = db = None
    nbytes = 0
    ws = None
    if need_dw:
        nbytes = _lib.load().cy_head1x1_bwd_ws_bytes(npix, Cc, K)
        ws = _ws(nbytes, x.device)
        if dw_into is not None:
            with ordered(("head_grad", dw_into.data_ptr())):
                _lib.call("cy_head1x1_bwd_into", x.data_ptr(), w2.data_ptr(), dlogits.data_ptr(), _ptr(dx),
                          dw_into.data_ptr(), _ptr(db_into), npix, Cc, K, dtype_code(x.dtype), _ptr(ws), nbytes, _stream())
            return dx, None, None
        dw = _f32(K * Cc, x.device)
        db = _f32(K, x.device)
    _lib.call("cy_head1x1_bwd", x.data_ptr(), w2.data_ptr(), dlogits.data_ptr(), _ptr(dx), _ptr(dw),
              _ptr(db), npix, Cc, K, dtype_code(x.dtype), _ptr(ws), nbytes, _stream())
    if dw is not None:
        dw = dw.view(K, Cc, 1, 1)
    return dx, dw, db


def softmax_kl_fwd(logits: Tensor, target: Tensor, eps: float) -> Tensor:
    N, K, H, W = logits.shape
    npix = N * H * W
    nbytes = _lib.load().cy_softmax_kl_ws_bytes(npix)
    ws = _ws(nbytes, logits.device)
    loss = _f32(1, logits.device)
    _lib.call("cy_softmax_kl_fwd", logits.data_ptr(), target.data_ptr(), loss.data_ptr(), npix, K,
              float(eps), ws.data_ptr(), nbytes, _stream())
    return loss.view(())


def softmax_kl_bwd(logits: Tensor, target: Tensor, gscale: Tensor, eps: float) -> Tensor:
    N, K, H, W = logits.shape
    d = empty_nhwc(N, K, H, W, torch.float32, logits.device)
    _lib.call("cy_softmax_kl_bwd", logits.data_ptr(), target.data_ptr(), gscale.data_ptr(),
              d.data_ptr(), N * H * W, K, float(eps), _stream())
    return d


def softmax_mse_fwd(a: Tensor, b: Tensor) -> Tensor:
    N, K, H, W = a.shape
    npix = N * H * W
    nbytes = _lib.load().cy_softmax_mse_ws_bytes(npix)
    ws = _ws(nbytes, a.device)
    loss = _f32(1, a.device)
    _lib.call("cy_softmax_mse_fwd", a.data_ptr(), b.data_ptr(), loss.data_ptr(), npix, K,
              ws.data_ptr(), nbytes, _stream())
    return loss.view(())


def softmax_mse_bwd(a: Tensor, b: Tensor, gscale: Tensor, need_a: bool, need_b: bool):
    N, K, H, W = a.shape
    da = empty_nhwc(N, K, H, W, torch.float32, a.device) if need_a else None
    db = empty_nhwc(N, K, H, W, torch.float32, a.device) if need_b else None
    _lib.call("cy_softmax_mse_bwd", a.data_ptr(), b.data_ptr(), gscale.data_ptr(), _ptr(da), _ptr(db),
              N * H * W, K, _stream())
    return da, db


def dice_counts(logits: Tensor, target: Tensor) -> Tensor:
    """int64 [N,K,2] = per-sample per-class (intersection, union) of argmax(logits) vs target."""
    N, K, H, W = logits.shape
    counts = torch.empty((N, K, 2), dtype=torch.int64, device=logits.device)
    _lib.call("cy_dice_counts", logits.data_ptr(), target.data_ptr(), counts.data_ptr(), N, H * W, K,
              _stream())
    return counts


# --------------------------------------------------------------------------- projector pieces
def avgpool_fwd(x: Tensor) -> Tensor:
    N, Cc, H, W = x.shape
    pooled = _f32(N * Cc, x.device).view(N, Cc)
    _lib.call("cy_avgpool_fwd", x.data_ptr(), pooled.data_ptr(), N, H * W, Cc, dtype_code(x.dtype),
              _stream())
    return pooled


def avgpool_bwd(dpooled: Tensor, shape, dtype) -> Tensor:
    N, Cc, H, W = shape
    dx = empty_nhwc(N, Cc, H, W, dtype, dpooled.device)
    _lib.call("cy_avgpool_bwd", dpooled.data_ptr(), dx.data_ptr(), N, H * W, Cc, dtype_code(dtype),
              _stream())
    return dx


def linear_fwd(x: Tensor, w: Tensor, b: Optional[Tensor], act: int = 0, slope: float = 0.01) -> Tensor:
    M, I = x.shape
    O = w.shape[0]
    y = _f32(M * O, x.device).view(M, O)
    _lib.call("cy_linear_fwd", x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), M, I, O, act,
              float(slope), _stream())
    return y


def linear_bwd(x: Tensor, w: Tensor, y: Tensor, dy: Tensor, act: int, slope: float, need_dx: bool,
               need_dw: bool, dw_into: Optional[Tensor] = None, db_into: Optional[Tensor] = None):
    """dx, dw, db; with dw_into / db_into the parameter gradients are ADDED into those buffers (returned as None)"""
    M, I = x.shape
    O = w.shape[0]
    dx = _f32(M * I, x.device).view(M, I) if need_dx else None
    if dw_into is not None:
        with ordered(("lin_grad", dw_into.data_ptr())):
            _lib.call("cy_linear_bwd_into", x.data_ptr(), w.data_ptr(), y.data_ptr(), dy.data_ptr(), _ptr(dx),
                      dw_into.data_ptr(), _ptr(db_into), M, I, O, act, float(slope), _stream())
        return dx, None, None
    dw = _f32(O * I, x.device).view(O, I) if need_dw else None
    db = _f32(O, x.device) if need_dw else None
    _lib.call("cy_linear_bwd", x.data_ptr(), w.data_ptr(), y.data_ptr(), dy.data_ptr(), _ptr(dx),
              _ptr(dw), _ptr(db), M, I, O, act, float(slope), _stream())
    return dx, dw, db


def proj_head_ok(x: Tensor, w1: Tensor, w2: Tensor) -> bool:
    """shapes the one-launch projection head takes (csrc/cy_contrast.hip proj_head_*)"""
    Cc, hid, out = x.shape[1], w1.shape[0], w2.shape[0]
    return (x.is_cuda and Cc % 8 == 0 and Cc <= 512 and hid % 4 == 0 and hid <= 512 and out <= 512
            and x.dtype in (torch.bfloat16, torch.float16, torch.float32))


def proj_head_fwd(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, slope: float = 0.01,
                  eps: float = 1e-12):
    """x NHWC [B,C,H,W] -> z [B,out] normalised, and (pooled, y1, y2, norms) for the backward"""
    B, Cc, H, W = x.shape
    hid, out = w1.shape[0], w2.shape[0]
    dev = x.device
    pooled, y1 = _f32(B * Cc, dev).view(B, Cc), _f32(B * hid, dev).view(B, hid)
    y2, norms = _f32(B * out, dev).view(B, out), _f32(B, dev)
    z = torch.empty((B, out), dtype=torch.float32, device=dev)  # (not a view: SupConLoss1 rejoins its two chunks through z)
    _lib.call("cy_proj_head_fwd", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
              pooled.data_ptr(), y1.data_ptr(), y2.data_ptr(), z.data_ptr(), norms.data_ptr(), B, H * W, Cc, hid, out,
              float(slope), float(eps), dtype_code(x.dtype), _stream())
    return z, pooled, y1, y2, norms


def proj_head_bwd(dz: Tensor, pooled: Tensor, y1: Tensor, y2: Tensor, norms: Tensor, w1: Tensor, w2: Tensor, shape,
                  dtype, need_dx: bool, sinks=None, slope: float = 0.01, eps: float = 1e-12):
    """-> dx (NHWC, dtype) or None, and (dw1, db1, dw2, db2) -- None each when `sinks` (the parameters' .grad buffers,
    added into) is given"""
    B, Cc, H, W = shape
    hid, out = w1.shape[0], w2.shape[0]
    dev = dz.device
    dx = empty_nhwc(B, Cc, H, W, dtype, dev) if need_dx else None
    nbytes = _lib.load().cy_proj_head_bwd_ws_bytes(B, hid, out)
    ws = _ws(nbytes, dev)
    if sinks is not None:
        g = sinks
        acc = 1
    else:
        g = (_f32(hid * Cc, dev).view(hid, Cc), _f32(hid, dev), _f32(out * hid, dev).view(out, hid), _f32(out, dev))
        acc = 0

    def launch():
        _lib.call("cy_proj_head_bwd", dz.data_ptr(), pooled.data_ptr(), y1.data_ptr(), y2.data_ptr(), norms.data_ptr(),
                  w1.data_ptr(), w2.data_ptr(), _ptr(dx), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                  g[3].data_ptr(), acc, ws.data_ptr(), nbytes, B, H * W, Cc, hid, out, float(slope), float(eps),
                  dtype_code(dtype), _stream())

    if sinks is not None:
        with ordered(("lin_grad", g[0].data_ptr())):
            launch()
        return dx, (None, None, None, None)
    launch()
    return dx, g


def l2norm_fwd(x: Tensor, eps: float = 1e-12):
    M, D = x.shape
    z = torch.empty_like(x)
    norms = _f32(M, x.device)
    _lib.call("cy_l2norm_fwd", x.data_ptr(), z.data_ptr(), norms.data_ptr(), M, D, float(eps), _stream())
    return z, norms


def l2norm_bwd(x: Tensor, norms: Tensor, dz: Tensor, eps: float = 1e-12) -> Tensor:
    M, D = x.shape
    dx = torch.empty_like(x)
    _lib.call("cy_l2norm_bwd", x.data_ptr(), norms.data_ptr(), dz.data_ptr(), dx.data_ptr(), M, D,
              float(eps), _stream())
    return dx


# --------------------------------------------------------------------------- SupCon / InfoNCE
def supcon_fwd(P: Tensor, labels: Optional[Tensor], pos_mask: Optional[Tensor], t: float):
    R, D = P.shape
    n = R // 2
    S = _f32(R * R, P.device).view(R, R)
    stats = _f32(R * 4, P.device).view(R, 4)
    loss = _f32(1, P.device)
    _lib.call("cy_supcon_fwd", P.data_ptr(), _ptr(labels), _ptr(pos_mask), S.data_ptr(),
              loss.data_ptr(), stats.data_ptr(), n, D, float(t), _stream())
    return loss.view(()), S, stats


def supcon_bwd(P: Tensor, labels, pos_mask, S: Tensor, stats: Tensor, gscale: Tensor, t: float) -> Tensor:
    R, D = P.shape
    G = _f32(R * R, P.device)
    dP = torch.empty_like(P)
    _lib.call("cy_supcon_bwd", P.data_ptr(), _ptr(labels), _ptr(pos_mask), S.data_ptr(),
              stats.data_ptr(), gscale.data_ptr(), G.data_ptr(), dP.data_ptr(), R // 2, D, float(t),
              _stream())
    return dP


def supcon_excl_fwd(P: Tensor, labels: Optional[Tensor], pos_mask: Optional[Tensor], t: float):
    """SupConLoss1(exclude_other_pos=True): (loss, S, row statistics, tmp) on the materialised similarity matrix"""
    R, D = P.shape
    S = _f32(R * R, P.device).view(R, R)
    stats = _f32(R * 4, P.device).view(R, 4)
    tmp = _f32(R * 4 + 1, P.device)
    loss = _f32(1, P.device)
    _lib.call("cy_supcon_excl_fwd", P.data_ptr(), _ptr(labels), _ptr(pos_mask), S.data_ptr(), loss.data_ptr(),
              stats.data_ptr(), tmp.data_ptr(), R // 2, D, float(t), _stream())
    return loss.view(()), S, stats, tmp


def supcon_excl_bwd(P: Tensor, labels, pos_mask, S: Tensor, stats: Tensor, tmp: Tensor, gscale: Tensor, t: float) -> Tensor:
    R, D = P.shape
    G = _f32(R * R, P.device)
    dP = torch.empty_like(P)
    _lib.call("cy_supcon_excl_bwd", P.data_ptr(), _ptr(labels), _ptr(pos_mask), S.data_ptr(), stats.data_ptr(),
              tmp.data_ptr(), gscale.data_ptr(), G.data_ptr(), dP.data_ptr(), R // 2, D, float(t), _stream())
    return dP


SUPCON_FUSED = os.environ.get("CY_SUPCON_FUSED", "1") != "0"


def supcon_fused_ok(P: Tensor) -> bool:
    return SUPCON_FUSED and P.shape[1] <= 256 and P.shape[1] % 8 == 0


def supcon_fwd_fused(P: Tensor, labels: Optional[Tensor], pos_mask: Optional[Tensor], t: float):
    """loss, diag(S), row statistics; the similarity matrix is never materialised (csrc/cy_contrast.hip)"""
    R, D = P.shape
    n = R // 2
    nbytes = _lib.load().cy_supcon_fused_ws_bytes(n, D)
    ws = _ws(nbytes, P.device)
    stats = _f32(R * 4, P.device).view(R, 4)
    diag = _f32(R, P.device)
    loss = _f32(1, P.device)
    _lib.call("cy_supcon_fused_fwd", P.data_ptr(), _ptr(labels), _ptr(pos_mask), loss.data_ptr(), stats.data_ptr(),
              diag.data_ptr(), ws.data_ptr(), nbytes, n, D, float(t), _stream())
    return loss.view(()), diag, stats


def supcon_bwd_fused(P: Tensor, labels, pos_mask, stats: Tensor, gscale: Tensor, t: float) -> Tensor:
    R, D = P.shape
    nbytes = _lib.load().cy_supcon_fused_ws_bytes(R // 2, D)
    ws = _ws(nbytes, P.device)
    dP = torch.empty_like(P)
    _lib.call("cy_supcon_fused_bwd", P.data_ptr(), _ptr(labels), _ptr(pos_mask), stats.data_ptr(), gscale.data_ptr(),
              dP.data_ptr(), ws.data_ptr(), nbytes, R // 2, D, float(t), _stream())
    return dP


def supcon_matrices(S: Tensor, stats: Tensor, labels, pos_mask):
    R = S.shape[0]
    outs = [_f32(R * R, S.device).view(R, R) for _ in range(4)]
    _lib.call("cy_supcon_matrices", S.data_ptr(), stats.data_ptr(), _ptr(labels), _ptr(pos_mask),
              *[o.data_ptr() for o in outs], R // 2, _stream())
    return outs  # sim_logits, sim_exp, pos_mask, neg_mask


def sgemm(A: Tensor, B: Tensor, alpha: float = 1.0, b_trans: bool = False) -> Tensor:
    M, K = A.shape
    N = B.shape[0] if b_trans else B.shape[1]
    Cm = _f32(M * N, A.device).view(M, N)
    _lib.call("cy_sgemm", A.data_ptr(), B.data_ptr(), Cm.data_ptr(), M, N, K, float(alpha),
              int(b_trans), _stream())
    return Cm


# --------------------------------------------------------------------------- affine / EMA / RAdam
def affine_fwd(x: Tensor, theta: Tensor, gamma: Optional[Tensor] = None) -> Tensor:
    require_gpu(x, theta)
    x = to_nhwc(x)
    N, Cc, H, W = x.shape
    out = empty_nhwc(N, Cc, H, W, x.dtype, x.device)
    _lib.call("cy_affine_nearest_fwd", x.data_ptr(), out.data_ptr(), theta.data_ptr(), _ptr(gamma), N,
              Cc, H, W, dtype_code(x.dtype), _stream())
    return out


def affine_bwd(dout: Tensor, theta: Tensor) -> Tensor:
    dout = to_nhwc(dout)
    N, Cc, H, W = dout.shape
    dx = empty_nhwc(N, Cc, H, W, dout.dtype, dout.device)
    _lib.call("cy_affine_nearest_bwd", dout.data_ptr(), dx.data_ptr(), theta.data_ptr(), N, Cc, H, W,
              dtype_code(dout.dtype), _stream())
    return dx


def ema_update(teacher: Tensor, student: Tensor, alpha: float, weight_decay: float) -> None:
    require_gpu(teacher, student)
    _lib.call("cy_ema_update", teacher.data_ptr(), student.data_ptr(), teacher.numel(), float(alpha),
              float(weight_decay), _stream())


def radam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, beta1: float, beta2: float,
               eps: float, wd: float, step: int) -> None:
    require_gpu(p, g)
    _lib.call("cy_radam_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
              float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step), _stream())


# --------------------------------------------------------------------------- dense projector / sampling
def _bins_tensor(bins, device) -> Optional[Tensor]:
    """host list/array of (image, bin row, bin col) -> int32 device tensor [nb,3] (sync-free upload)"""
    if bins is None:
        return None
    if isinstance(bins, Tensor):
        return bins if bins.is_cuda else pinned.upload(bins.to(torch.int32).contiguous(), device)
    return pinned.upload(torch.as_tensor(bins, dtype=torch.int32).reshape(-1, 3).contiguous(), device)


def dense_proj_fwd(x: Tensor, w1: Tensor, b1: Tensor, size: Tuple[int, int], bins: Optional[Tensor],
                   slope: float = 0.01) -> Tensor:
    """x NHWC [N,C,H,W]; w1 f32 [hid,C]; -> hpool f32 [nb, hid] = mean over each bin of lrelu(w1 x + b1)"""
    require_gpu(x, w1, b1)
    N, Cc, H, W = x.shape
    hid = w1.shape[0]
    nb = N * size[0] * size[1] if bins is None else bins.shape[0]
    out = _f32(nb * hid, x.device).view(nb, hid)
    _lib.call("cy_dense_proj_fwd", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), _ptr(bins), nb,
              out.data_ptr(), N, H, W, Cc, Cc, hid, size[0], size[1], slope, dtype_code(x.dtype), _stream())
    return out


def dense_proj_bwd(x: Tensor, w1: Tensor, b1: Tensor, size: Tuple[int, int], bins: Optional[Tensor],
                   dhpool: Tensor, need_dx: bool, need_dw: bool, slope: float = 0.01):
    N, Cc, H, W = x.shape
    hid = w1.shape[0]
    nb = dhpool.shape[0]
    dx = None
    if need_dx:
        dx = torch.zeros((N, H, W, Cc), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
    dw = _f32(hid * Cc, x.device).view(hid, Cc) if need_dw else None
    db = _f32(hid, x.device) if need_dw else None
    nbytes = _lib.load().cy_dense_proj_bwd_ws_bytes(nb, Cc, hid)
    ws = _ws(nbytes, x.device)
    _lib.call("cy_dense_proj_bwd", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), _ptr(bins), nb,
              dhpool.data_ptr(), _ptr(dx), _ptr(dw), _ptr(db), 0, N, H, W, Cc, Cc, hid, size[0], size[1],
              slope, dtype_code(x.dtype), ws.data_ptr(), nbytes, _stream())
    return dx, dw, db


def adaptive_avgpool_fwd(x: Tensor, size: Tuple[int, int], bins: Optional[Tensor] = None) -> Tensor:
    require_gpu(x)
    N, Cc, H, W = x.shape
    nb = N * size[0] * size[1] if bins is None else bins.shape[0]
    out = _f32(nb * Cc, x.device).view(nb, Cc)
    _lib.call("cy_adaptive_avgpool_fwd", x.data_ptr(), _ptr(bins), nb, out.data_ptr(), N, H, W, Cc, Cc,
              size[0], size[1], dtype_code(x.dtype), _stream())
    return out


def adaptive_avgpool_bwd(dpool: Tensor, shape, dtype, size: Tuple[int, int]) -> Tensor:
    N, Cc, H, W = shape
    dx = empty_nhwc(N, Cc, H, W, dtype, dpool.device)
    _lib.call("cy_adaptive_avgpool_bwd", dpool.data_ptr(), dx.data_ptr(), N, H, W, Cc, Cc, size[0], size[1],
              dtype_code(dtype), _stream())
    return dx


def adaptive_maxpool_fwd(x: Tensor, size: Tuple[int, int]):
    """nn.AdaptiveMaxPool2d(size) on an NHWC map -> (rows [N*sh*sw, C] f32, arg-max pixel indices int32)"""
    require_gpu(x)
    N, Cc, H, W = x.shape
    nb = N * size[0] * size[1]
    out = _f32(nb * Cc, x.device).view(nb, Cc)
    arg = torch.empty((nb, Cc), dtype=torch.int32, device=x.device)
    _lib.call("cy_adaptive_maxpool_fwd", x.data_ptr(), out.data_ptr(), arg.data_ptr(), N, H, W, Cc, Cc, size[0], size[1],
              dtype_code(x.dtype), _stream())
    return out, arg


def adaptive_maxpool_bwd(dpool: Tensor, arg: Tensor, shape, dtype, size: Tuple[int, int]) -> Tensor:
    N, Cc, H, W = shape
    dx = empty_nhwc(N, Cc, H, W, dtype, dpool.device)
    _lib.call("cy_adaptive_maxpool_bwd", dpool.data_ptr(), arg.data_ptr(), dx.data_ptr(), N, H, W, Cc, Cc, size[0],
              size[1], dtype_code(dtype), _stream())
    return dx


def gather_rows_fwd(src: Tensor, idx: Tensor) -> Tensor:
    require_gpu(src, idx)
    M, D = idx.shape[0], src.shape[1]
    out = _f32(M * D, src.device).view(M, D)
    _lib.call("cy_gather_rows_fwd", src.data_ptr(), idx.data_ptr(), out.data_ptr(), M, D, _stream())
    return out


def gather_rows_bwd(dout: Tensor, idx: Tensor, rows: int) -> Tensor:
    M, D = dout.shape
    dsrc = torch.zeros((rows, D), dtype=torch.float32, device=dout.device)
    _lib.call("cy_gather_rows_bwd", dout.data_ptr(), idx.data_ptr(), dsrc.data_ptr(), M, D, _stream())
    return dsrc


# --------------------------------------------------------------------------- cluster heads / discrete MI
def cluster_head_ok(C: int, S: int, k: int) -> bool:
    return C in (32, 64) and S * k <= 128 and k <= 32


def cluster_head_fwd(x: Tensor, w: Tensor, b: Optional[Tensor], S: int, k: int, T: float = 1.0) -> Tensor:
    """x: [M, C] rows (an NHWC map flattened over pixels) -> probs f32 [S, M, k] (csrc/cy_cluster_head.hip)"""
    require_gpu(x, w)
    M, Cc = x.shape
    probs = _f32(S * M * k, x.device).view(S, M, k)
    _lib.call("cy_cluster_head_fwd", x.data_ptr(), w.data_ptr(), _ptr(b), probs.data_ptr(), M, Cc, S * k, S, k,
              1.0 / T, dtype_code(x.dtype), _stream())
    return probs


def cluster_head_bwd(x: Tensor, w: Tensor, probs: Tensor, dprobs: Tensor, T: float, need_dx: bool, need_dw: bool):
    S, M, k = probs.shape
    Cc = x.shape[1]
    dx = torch.empty_like(x) if need_dx else None
    dw = db = ws = None
    nbytes = 0
    if need_dw:
        dw, db = _f32(S * k * Cc, x.device).view(S * k, Cc), _f32(S * k, x.device)
        nbytes = _lib.load().cy_cluster_head_bwd_ws_bytes(M, Cc)
        ws = _ws(nbytes, x.device)
    _lib.call("cy_cluster_head_bwd", x.data_ptr(), w.data_ptr(), probs.data_ptr(), dprobs.data_ptr(), _ptr(dx),
              _ptr(dw), _ptr(db), M, Cc, S * k, S, k, 1.0 / T, dtype_code(x.dtype), _ptr(ws), nbytes, _stream())
    return dx, dw, db


def group_softmax_fwd(logits: Tensor, S: int, k: int, T: float = 1.0) -> Tensor:
    """logits f32 [M, S*k] -> probs f32 [S, M, k]"""
    require_gpu(logits)
    M = logits.shape[0]
    probs = _f32(S * M * k, logits.device).view(S, M, k)
    _lib.call("cy_group_softmax_fwd", logits.data_ptr(), probs.data_ptr(), M, S, k, 1.0 / T, _stream())
    return probs


def group_softmax_bwd(probs: Tensor, dprobs: Tensor, T: float = 1.0) -> Tensor:
    S, M, k = probs.shape
    dl = _f32(M * S * k, probs.device).view(M, S * k)
    _lib.call("cy_group_softmax_bwd", probs.data_ptr(), dprobs.data_ptr(), dl.data_ptr(), M, S, k, 1.0 / T,
              _stream())
    return dl


def joint_fwd(x1: Tensor, x2: Tensor, N: int, H: int, W: int, k: int, pad: int, normalise: bool) -> Tensor:
    """x1, x2: f32 NHWC-contiguous [N,H,W,k] buffers -> J f32 [T*T, k, k]"""
    require_gpu(x1, x2)
    T = 2 * pad + 1
    J = _f32(T * T * k * k, x1.device).view(T * T, k, k)
    nbytes = _lib.load().cy_joint_ws_bytes(N, H, W, k, pad)
    ws = _ws(nbytes, x1.device)
    _lib.call("cy_joint_fwd", x1.data_ptr(), x2.data_ptr(), J.data_ptr(), N, H, W, k, pad, int(normalise),
              ws.data_ptr(), nbytes, _stream())
    return J


def joint_bwd(x1: Tensor, x2: Tensor, dJ: Tensor, gscale: Tensor, N: int, H: int, W: int, k: int, pad: int,
              normalise: bool, need1: bool, need2: bool):
    d1 = torch.empty_like(x1) if need1 else None
    d2 = torch.empty_like(x2) if need2 else None
    _lib.call("cy_joint_bwd", x1.data_ptr(), x2.data_ptr(), dJ.data_ptr(), gscale.data_ptr(), _ptr(d1), _ptr(d2),
              N, H, W, k, pad, int(normalise), _stream())
    return d1, d2


def iid_loss(J: Tensor, mode: int, symmetric: bool, lamda: float, eps: float, want_grad: bool = True):
    """-> (out2 [2] = loss(lamda), loss(1); P [TT,k,k] normalised joint; dJ or None)"""
    TT, k = J.shape[0], J.shape[1]
    out2 = _f32(2, J.device)
    P = torch.empty_like(J)
    dJ = torch.empty_like(J) if want_grad else None
    nbytes = _lib.load().cy_iid_loss_ws_bytes(TT, k)
    ws = _ws(nbytes, J.device)
    _lib.call("cy_iid_loss", J.data_ptr(), out2.data_ptr(), P.data_ptr(), _ptr(dJ), TT, k, mode, int(symmetric),
              lamda, eps, ws.data_ptr(), nbytes, _stream())
    return out2, P, dJ


# --------------------------------------------------------------------------- GroupNorm + SiLU, bilinear
def gn_silu_fwd(y: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, groups: int, eps: float):
    require_gpu(y, gamma, beta)
    N, Cc, H, W = y.shape
    out = empty_nhwc(N, Cc, H, W, y.dtype, y.device)
    mr = _f32(N * groups * 2, y.device)
    nbytes = _lib.load().cy_gn_ws_bytes(N, Cc)
    ws = _ws(nbytes, y.device)
    _lib.call("cy_gn_silu_fwd", y.data_ptr(), Cc, _ptr(bias), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
              Cc, mr.data_ptr(), N, H * W, Cc, groups, eps, dtype_code(y.dtype), ws.data_ptr(), nbytes, _stream())
    return out, mr


def gn_silu_bwd(y: Tensor, dz: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, mr: Tensor,
                groups: int):
    N, Cc, H, W = y.shape
    du = empty_nhwc(N, Cc, H, W, y.dtype, y.device)
    dg, dbt = _f32(Cc, y.device), _f32(Cc, y.device)
    dbias = _f32(Cc, y.device) if bias is not None else None
    nbytes = _lib.load().cy_gn_ws_bytes(N, Cc)
    ws = _ws(nbytes, y.device)
    _lib.call("cy_gn_silu_bwd", y.data_ptr(), Cc, dz.data_ptr(), Cc, _ptr(bias), gamma.data_ptr(), beta.data_ptr(),
              mr.data_ptr(), du.data_ptr(), Cc, dg.data_ptr(), dbt.data_ptr(), _ptr(dbias), 0, N, H * W, Cc, groups,
              dtype_code(y.dtype), ws.data_ptr(), nbytes, _stream())
    return du, dg, dbt, dbias


def gn_silu_mod_fwd(y: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, mod_s: Tensor, mod_t: Tensor,
                    groups: int, eps: float):
    """gn_silu_fwd with the time-embedding modulation (x * (scale + 1) + shift between the norm and SiLU,
    contrastyou/arch/unet2.py:216-220); mod_s / mod_t: f32 [N, C]"""
    require_gpu(y, gamma, beta, mod_s, mod_t)
    N, Cc, H, W = y.shape
    out = empty_nhwc(N, Cc, H, W, y.dtype, y.device)
    mr = _f32(N * groups * 2, y.device)
    nbytes = _lib.load().cy_gn_ws_bytes(N, Cc)
    ws = _ws(nbytes, y.device)
    _lib.call("cy_gn_silu_mod_fwd", y.data_ptr(), Cc, _ptr(bias), gamma.data_ptr(), beta.data_ptr(), mod_s.data_ptr(),
              mod_t.data_ptr(), out.data_ptr(), Cc, mr.data_ptr(), N, H * W, Cc, groups, eps, dtype_code(y.dtype),
              ws.data_ptr(), nbytes, _stream())
    return out, mr


def gn_silu_mod_bwd(y: Tensor, dz: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, mod_s: Tensor,
                    mod_t: Tensor, mr: Tensor, groups: int):
    N, Cc, H, W = y.shape
    du = empty_nhwc(N, Cc, H, W, y.dtype, y.device)
    dg, dbt = _f32(Cc, y.device), _f32(Cc, y.device)
    dbias = _f32(Cc, y.device) if bias is not None else None
    dms, dmt = _f32(N * Cc, y.device).view(N, Cc), _f32(N * Cc, y.device).view(N, Cc)
    nbytes = _lib.load().cy_gn_ws_bytes(N, Cc)
    ws = _ws(nbytes, y.device)
    _lib.call("cy_gn_silu_mod_bwd", y.data_ptr(), Cc, dz.data_ptr(), Cc, _ptr(bias), gamma.data_ptr(), beta.data_ptr(),
              mod_s.data_ptr(), mod_t.data_ptr(), mr.data_ptr(), du.data_ptr(), Cc, dg.data_ptr(), dbt.data_ptr(),
              _ptr(dbias), dms.data_ptr(), dmt.data_ptr(), 0, N, H * W, Cc, groups, dtype_code(y.dtype), ws.data_ptr(),
              nbytes, _stream())
    return du, dg, dbt, dbias, dms, dmt


ACT_SILU, ACT_GELU = 0, 1


def act_fwd(x: Tensor, kind: int) -> Tensor:
    """elementwise SiLU / exact GELU of a small f32 tensor (the time-embedding MLPs of UNet2)"""
    require_gpu(x)
    y = torch.empty_like(x)
    _lib.call("cy_act_fwd", x.data_ptr(), y.data_ptr(), x.numel(), kind, _stream())
    return y


def act_bwd(x: Tensor, dy: Tensor, kind: int) -> Tensor:
    dx = torch.empty_like(x)
    _lib.call("cy_act_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), kind, _stream())
    return dx


def sinusoidal_emb(time: Tensor, dim: int) -> Tensor:
    """SinusoidalPosEmb(dim)(time) (contrastyou/arch/unet2.py:161-173): [B] -> [B, dim]"""
    require_gpu(time)
    t = time.detach().float().contiguous()
    out = _f32(t.numel() * dim, t.device).view(t.numel(), dim)
    _lib.call("cy_sinusoidal_emb", t.data_ptr(), out.data_ptr(), t.numel(), dim, _stream())
    return out


def bilinear_fwd(x: Tensor, size: Tuple[int, int]) -> Tensor:
    require_gpu(x)
    x = to_nhwc(x)
    N, Cc, H, W = x.shape
    out = empty_nhwc(N, Cc, size[0], size[1], x.dtype, x.device)
    _lib.call("cy_bilinear_fwd", x.data_ptr(), out.data_ptr(), N, H, W, Cc, size[0], size[1], dtype_code(x.dtype),
              _stream())
    return out
