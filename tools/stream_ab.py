#!/usr/bin/env python3
"""The streaming conv kernel's four forms on the 224 x 224, 32 -> 32 layer (Conv1b / Up_conv2b): BN+ReLU prologue
on / off x statistics epilogue on / off.  Run on the GPU box:  python tools/stream_ab.py [--n 32]"""
import argparse
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops  # noqa: E402
from bench_layers import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--c", type=int, default=32)
    ap.add_argument("--hw", type=int, default=224)
    a = ap.parse_args()
    dt, dev = torch.bfloat16, "cuda"
    N, C, H = a.n, a.c, a.hw
    x = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    wf, _ = ops.pack_weights(w, dt)
    scale, shift = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
    nbytes = 2 * x.numel() * 2
    print(ops.conv3x3_plan(N, H, H, C, 0, C, dt, 0, True))
    for pro in (False, True):
        for stats in (False, True):
            t = timeit(lambda: ops.conv3x3_fwd(x, None, wf, C, scale=scale if pro else None, shift=shift if pro else None,
                                               want_stats=stats), 20, 3)
            print(f"prologue {int(pro)} statistics {int(stats)}: {t * 1e3:7.1f} us  {nbytes / t / 1e9:7.1f} TB/s (in + out once)")


if __name__ == "__main__":
    main()
