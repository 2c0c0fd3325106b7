#!/usr/bin/env python3
"""Per-layer timing of the U-Net conv kernels at a BASELINE config (default C2: N=32, 224^2,
max_channel 512, bf16): forward, data-gradient and weight-gradient kernels, in TFLOP/s.
Run on the GPU box:  python tools/bench_layers.py [--n 32] [--hw 224] [--dtype bf16]"""
import argparse
import os
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--maxc", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="", help="comma list of layer names")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pool-on-load", action="store_true",
                    help="Conv(i)a with the 2x2 max on load (the U-Net reads the pooled tensor bn_relu_apply_pool writes)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda"
    c = [a.maxc // 16 * m for m in (1, 2, 4, 8, 16)]
    hw = [a.hw // (2 ** i) for i in range(5)]
    # (name, H, C1, C2, Cout, mode, prologue)
    layers = [("Conv1b", hw[0], c[0], 0, c[0], 0, 1)]
    for i in range(1, 5):
        layers.append((f"Conv{i+1}a", hw[i], c[i - 1], 0, c[i], 1 if a.pool_on_load else 0, 0))
        layers.append((f"Conv{i+1}b", hw[i], c[i], 0, c[i], 0, 1))
    for i in range(3, -1, -1):
        layers.append((f"Up{i+2}", hw[i], c[i + 1], 0, c[i], 2, 0))
        layers.append((f"Up_conv{i+2}a", hw[i], c[i], c[i], c[i], 0, 0))
        layers.append((f"Up_conv{i+2}b", hw[i], c[i], 0, c[i], 0, 1))
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totf = 0.0
    print(f"{'layer':12s} {'HxW':>5s} {'Cin':>5s} {'Cout':>5s} | {'fwd ms':>8s} {'TF/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
    only = set(x for x in a.only.split(",") if x)
    for name, H, C1, C2, Cout, mode, pro in layers:
        if only and name not in only:
            continue
        N = a.n
        sh = 2 * H if mode == 1 else (H // 2 if mode == 2 else H)
        x1 = ops.empty_nhwc(N, C1, sh, sh, dt, dev).normal_()
        x2 = ops.empty_nhwc(N, C2, H, H, dt, dev).normal_() if C2 else None
        w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.05
        wf, wd = ops.pack_weights(w, dt)
        scale = torch.rand(C1, device=dev) + 0.5 if pro else None
        shift = torch.rand(C1, device=dev) - 0.5 if pro else None
        dy = ops.empty_nhwc(N, Cout, H, H, dt, dev).normal_()
        flops = 2.0 * N * H * H * 9 * (C1 + C2) * Cout
        t_f = timeit(lambda: ops.conv3x3_fwd(x1, x2, wf, Cout, mode=mode, scale=scale, shift=shift), a.iters, 1)
        t_d = timeit(lambda: ops.conv3x3_fwd(dy, None, wd, C1 + C2, want_stats=False), a.iters, 1)
        t_w = timeit(lambda: ops.conv3x3_wgrad(x1, x2, dy, mode=mode, scale=scale, shift=shift), a.iters, 1)
        tot["fwd"] += t_f
        tot["dgrad"] += t_d
        tot["wgrad"] += t_w
        totf += flops
        pf_ = ops.conv3x3_plan(N, H, H, C1, C2, Cout, dt, mode, bool(pro))
        pd_ = ops.conv3x3_plan(N, H, H, Cout, 0, C1 + C2, dt, 0, False)
        tag = lambda q: f"{q['kernel'][8:-7]}:{q['th']}x{q['bn']}z{q['ksplit']}w{q['workgroups']}"  # noqa: E731
        print(f"{name:12s} {H:5d} {C1+C2:5d} {Cout:5d} | {t_f:8.3f} {flops/t_f/1e9:7.1f} | {t_d:8.3f} {flops/t_d/1e9:7.1f} | {t_w:8.3f} {flops/t_w/1e9:7.1f} | {tag(pf_)} {tag(pd_)}")
    for k, v in tot.items():
        print(f"total {k}: {v:.3f} ms  {totf/v/1e9:.1f} TFLOP/s")
    print(f"sum: {sum(tot.values()):.3f} ms for N={a.n}")


if __name__ == "__main__":
    main()
