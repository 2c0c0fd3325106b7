#!/usr/bin/env python3
"""Elementwise BatchNorm kernels, finalize path against accumulator path (csrc/cy_bn_acc.h), back-to-back launches:
    python tools/bn_fold_bench.py            (on the GPU box)"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    dev, dt = "cuda", torch.bfloat16
    print(f"{'shape':>22s} {'R':>3s} | {'apply':>7s} {'fold':>7s} | {'bwd red+fin+apply':>18s} {'red_acc+apply_fold':>19s} | {'bwd apply':>9s} {'apply_fold':>10s} us")
    for N, C, H in ((16, 32, 224), (32, 32, 224), (16, 64, 112), (16, 128, 56), (16, 256, 28), (16, 512, 14), (32, 512, 14)):
        y = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
        da = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
        gm, bt = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        wf, _ = ops.pack_weights(w, dt, want_dgrad=False)
        _, part = ops.conv3x3_fwd(y, None, wf, C)
        yy, acc = ops.conv3x3_fwd(y, None, wf, C, stats_acc=True)
        cnt = N * H * H
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        sc, sh, mean, istd = ops.bn_finalize(part, cnt, gm, bt, rm, rv, 0.1, 1e-5, True, False, C, dev)
        st = ops.BnState(acc, gm, bt, cnt, 1e-5, dev)
        t0 = timeit(lambda: ops.bn_relu_apply(yy, sc, sh))
        t1 = timeit(lambda: ops.bn_relu_apply_fold(yy, st))
        t2 = timeit(lambda: ops.bn_relu_bwd(da, yy, sc, sh, mean, istd, True))
        coef = st.coef

        def bwd_acc():
            a = ops.bn_acc_new(C, ops._lib.call("cy_bn_relu_bwd_workgroups", cnt, C), dev)
            ops.bn_relu_bwd_acc(da, yy, coef[0], True, acc=a)
        t3 = timeit(bwd_acc)
        a = ops.bn_acc_new(C, ops._lib.call("cy_bn_relu_bwd_workgroups", cnt, C), dev)
        ops.bn_relu_bwd_acc(da, yy, coef[0], True, acc=a)
        t5 = timeit(lambda: ops.bn_relu_bwd_acc(da, yy, coef[0], True, acc=a, acc_filled=True))
        kc = torch.zeros(2 * C, device=dev)
        dy = torch.empty_like(yy)
        t4 = timeit(lambda: ops._lib.call("cy_bn_relu_bwd_apply", da.data_ptr(), C, yy.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                          kc.data_ptr(), dy.data_ptr(), cnt, C, ops.dtype_code(dt), ops._stream()))
        print(f"{str((N, C, H, H)):>22s} {acc.R:3d} | {t0:7.1f} {t1:7.1f} | {t2:18.1f} {t3:19.1f} | {t4:9.1f} {t5:10.1f}")


if __name__ == "__main__":
    main()
