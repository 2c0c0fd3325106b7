#!/usr/bin/env python3
"""Data gradient with the backward sums of the next BatchNorm in its epilogue (cy_conv3x3_dgrad_dz) against the data
gradient + the reduce launch it replaces:   python tools/bench_dgrad_dz.py [--n 16]"""
import argparse
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops  # noqa: E402
from bench_dgrad_bn import timeit  # noqa: E402

ops.DGRAD_DZ = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    a = ap.parse_args()
    dev, dt, N = "cuda", torch.bfloat16, a.n
    c = [32, 64, 128, 256, 512]
    hw = [224, 112, 56, 28, 14]
    layers = []
    for i in range(1, 5):
        layers.append((f"Conv{i+1}b", hw[i], c[i], c[i], None))
    for i in range(3, -1, -1):
        layers.append((f"Up_conv{i+2}a", hw[i], c[i], 2 * c[i], c[i]))
        layers.append((f"Up_conv{i+2}b", hw[i], c[i], c[i], None))
    print(f"{'layer':12s} {'HxW':>4s} {'C':>4s} {'Cin':>4s} | {'dgrad':>7s} {'reduce':>7s} {'sum':>7s} | {'fused':>7s}")
    for name, H, C, Cin, split in layers:
        dy = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
        w = torch.randn(C, Cin, 3, 3, device=dev) * 0.05
        _, wd = ops.pack_weights(w, dt)
        Cs = Cin - split if split else Cin
        y = ops.empty_nhwc(N, Cs, H, H, dt, dev).normal_()
        coef = torch.rand(5, Cs, device=dev) + 0.5
        acc = ops.bn_bwd_acc_new(N, Cs, H, H, False, dev)
        ok = ops.conv3x3_dgrad_dz_ok(dy, Cin, split, split or 0, Cs)
        ref, _ = ops.conv3x3_fwd(dy, None, wd, Cin, want_stats=False, split=split)
        da = ref[1] if split else ref
        t_d = timeit(lambda: ops.conv3x3_fwd(dy, None, wd, Cin, want_stats=False, split=split))
        t_r = timeit(lambda: ops.bn_bwd_reduce_acc(da, y, coef[0], acc))
        t_f = timeit(lambda: ops.conv3x3_dgrad_dz(dy, wd, Cin, y, coef[0], acc, split=split)) if ok else float("nan")
        print(f"{name:12s} {H:4d} {C:4d} {Cin:4d} | {t_d:7.1f} {t_r:7.1f} {t_d + t_r:7.1f} | {t_f:7.1f}")


if __name__ == "__main__":
    sys.path.insert(0, str(REPO / "tools"))
    main()
