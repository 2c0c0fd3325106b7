#!/usr/bin/env python3
"""The BatchNorm-family elementwise kernels on COLD data: every launch of a timed run works on another set of tensors
(sets rotate over > 1 GB, well past the 256 MB Infinity Cache), like the launches of a training step, whose operands
were written a few hundred MB of traffic ago.  Prints us per launch and the HBM rate over the algorithmic bytes.
    python tools/bench_ew_cold.py            (on the GPU box)"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops  # noqa: E402
from cyhip import _lib  # noqa: E402


def timed(fns, rounds=6):
    """one pass over the sets, captured into a HIP graph (no Python between the launches), replayed"""
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(rounds):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (rounds * len(fns)) * 1e3


def main():
    dev, dt = "cuda", torch.bfloat16
    code = ops.dtype_code(dt)
    print(f"{'shape':>20s} {'MB':>6s} | {'apply':>12s} | {'apply+pool':>12s} | {'bwd sums':>12s} | {'bwd apply':>12s} | {'pool bwd+sums':>13s} | {'apply, coef':>12s} | {'bwd ap, coef':>12s}  us (TB/s)")
    for N, C, H in ((32, 32, 224), (16, 32, 224), (32, 64, 112), (16, 64, 112), (32, 128, 56), (16, 128, 56), (32, 256, 28)):
        unit = N * C * H * H * 2
        nset = max(2, int(1.3e9 // (3 * unit)) + 1)
        nset = min(nset, 24)
        npix = N * H * H
        sets = []
        gm, bt = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        wf, _ = ops.pack_weights(w, dt, want_dgrad=False)
        for _ in range(nset):
            x = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
            y, acc = ops.conv3x3_fwd(x, None, wf, C, stats_acc=True)
            del x
            bs = ops.BnState(acc, gm, bt, npix, 1e-5, dev)
            da = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
            out = ops.bn_relu_apply_fold(y, bs)
            pooled = ops.empty_nhwc(N, C, H // 2, H // 2, dt, dev)
            dpool = ops.empty_nhwc(N, C, H // 2, H // 2, dt, dev).normal_()
            bacc = ops.bn_bwd_acc_new(N, C, H, H, True, dev)
            ops.bn_bwd_reduce_acc(da, y, bs.coef, bacc)
            sets.append((y, bs, da, out, pooled, dpool, bacc))
        torch.cuda.synchronize()

        def f_apply(s):
            y, bs, da, out, pooled, dpool, bacc = s
            return lambda: _lib.call("cy_bn_relu_apply_fold", y.data_ptr(), bs.ref, out.data_ptr(), npix, code, code, ops._stream())

        def f_pool(s):
            y, bs, da, out, pooled, dpool, bacc = s
            return lambda: _lib.call("cy_bn_relu_apply_pool_fold", y.data_ptr(), bs.ref, out.data_ptr(), pooled.data_ptr(), N,
                                     H // 2, H // 2, code, code, ops._stream())

        def f_sums(s):
            y, bs, da, out, pooled, dpool, bacc = s
            return lambda: _lib.call("cy_bn_relu_bwd_reduce_acc", da.data_ptr(), C, y.data_ptr(), bs.coef.data_ptr(), bacc.ref,
                                     npix, C, code, ops._stream())

        def f_bapply(s):
            y, bs, da, out, pooled, dpool, bacc = s
            # (writes dy over `out`: same traffic, no further buffer)
            return lambda: _lib.call("cy_bn_relu_bwd_apply_fold", da.data_ptr(), C, y.data_ptr(), bs.coef.data_ptr(), bacc.ref,
                                     float(npix), 1, 0, 0, 0, out.data_ptr(), npix, C, code, ops._stream())

        kc = torch.zeros(2 * C, device=dev)

        def f_apply_plain(s):
            y, bs, da, out, pooled, dpool, bacc = s
            return lambda: _lib.call("cy_bn_relu_apply", y.data_ptr(), bs.coef[0].data_ptr(), bs.coef[1].data_ptr(), out.data_ptr(),
                                     npix, C, code, code, ops._stream())

        def f_bapply_plain(s):
            y, bs, da, out, pooled, dpool, bacc = s
            return lambda: _lib.call("cy_bn_relu_bwd_apply", da.data_ptr(), C, y.data_ptr(), bs.coef[0].data_ptr(),
                                     bs.coef[1].data_ptr(), kc.data_ptr(), out.data_ptr(), npix, C, code, ops._stream())

        def f_poolbwd(s):
            y, bs, da, out, pooled, dpool, bacc = s
            return lambda: ops.maxpool2_bwd_bn_acc(out, dpool, da, y, bs.coef[0], bacc)

        res = []
        for mk, units in ((f_apply, 2.0), (f_pool, 2.25), (f_sums, 2.0), (f_bapply, 3.0), (f_poolbwd, 3.25), (f_apply_plain, 2.0),
                          (f_bapply_plain, 3.0)):
            try:
                us = timed([mk(s) for s in sets])
                res.append(f"{us:6.1f} ({units * unit / us / 1e6:4.2f})")
            except Exception as ex:  # noqa: BLE001
                res.append(f"  n/a {type(ex).__name__[:6]}")
        print(f"{str((N, C, H, H)):>20s} {unit / 1e6:6.1f} | " + " | ".join(f"{r:>12s}" for r in res))
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
