#!/usr/bin/env python3
"""Data gradient of the U-Net's 3x3 layers with the BatchNorm + ReLU backward in the load path (cy_conv3x3_dgrad_bn)
against the two launches it replaces (cy_bn_relu_bwd_apply_fold + cy_conv3x3_fwd), per layer:
    python tools/bench_dgrad_bn.py [--n 16]"""
import argparse
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops  # noqa: E402

ops.DGRAD_BN = True


def timeit(fn, iters=20):
    for _ in range(2):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    a = ap.parse_args()
    dev, dt, N = "cuda", torch.bfloat16, a.n
    c = [32, 64, 128, 256, 512]
    hw = [224, 112, 56, 28, 14]
    # (name, H, C of dy (= the forward layer's Cout), Cin of the data gradient, split)
    layers = [("Conv1b", hw[0], c[0], c[0], None)]
    for i in range(1, 5):
        layers.append((f"Conv{i+1}a", hw[i], c[i], c[i - 1], None))
        layers.append((f"Conv{i+1}b", hw[i], c[i], c[i], None))
    for i in range(3, -1, -1):
        layers.append((f"Up{i+2}", hw[i], c[i], c[i + 1], None))
        layers.append((f"Up_conv{i+2}a", hw[i], c[i], 2 * c[i], c[i]))
        layers.append((f"Up_conv{i+2}b", hw[i], c[i], c[i], None))
    print(f"{'layer':12s} {'HxW':>4s} {'C':>4s} {'Cin':>4s} | {'apply':>7s} {'dgrad':>7s} {'sum':>7s} | {'fused':>7s} | plan")
    tot = [0.0, 0.0]
    for name, H, C, Cin, split in layers:
        y = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
        da = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
        w = torch.randn(C, Cin, 3, 3, device=dev) * 0.05
        _, wd = ops.pack_weights(w, dt)
        coef = torch.rand(5, C, device=dev) + 0.5
        acc = ops.bn_bwd_acc_new(N, C, H, H, False, dev)
        ops.bn_bwd_reduce_acc(da, y, coef[0], acc)
        dy, _, _ = ops.bn_relu_bwd_acc(da, y, coef[0], True, acc=acc, acc_filled=True)
        kc = torch.zeros(2 * C, device=dev)
        t_ap = timeit(lambda: ops._lib.call("cy_bn_relu_bwd_apply_fold", da.data_ptr(), C, y.data_ptr(), coef.data_ptr(), acc.ref,
                                            float(N * H * H), 1, None, None, 0, dy.data_ptr(), N * H * H, C, ops.dtype_code(dt), ops._stream()))
        t_dg = timeit(lambda: ops.conv3x3_fwd(dy, None, wd, Cin, want_stats=False, split=split))
        ok = ops.conv3x3_dgrad_bn_ok(da, Cin, split)
        t_fu = timeit(lambda: ops.conv3x3_dgrad_bn(da, y, coef[0], acc, True, wd, Cin, want_param_grads=False, split=split)) if ok else float("nan")
        p = ops.conv3x3_plan(N, H, H, C, 0, Cin, dt, 0, False)
        tag = f"{p['kernel'][8:-7]}:{p['th']}x{p['bn']}z{p['ksplit']}w{p['workgroups']}"
        print(f"{name:12s} {H:4d} {C:4d} {Cin:4d} | {t_ap:7.1f} {t_dg:7.1f} {t_ap + t_dg:7.1f} | {t_fu:7.1f} | {tag}")
        if ok:
            tot[0] += t_ap + t_dg
            tot[1] += t_fu
    print(f"fused layers: two launches {tot[0]:.0f} us, one launch {tot[1]:.0f} us")


if __name__ == "__main__":
    main()
