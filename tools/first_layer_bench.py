#!/usr/bin/env python3
"""First-layer kernels (1 -> 32 channels at 224 x 224) over batch sizes, MFMA form against the VALU form.
Run on the GPU box:  python tools/first_layer_bench.py"""
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def child():
    import torch
    sys.path.insert(0, str(REPO / "contrast-you_amd"))
    sys.path.insert(0, str(REPO / "tools"))
    from cyhip import ops
    from bench_layers import timeit
    w = torch.randn(32, 1, 3, 3, device="cuda") * 0.2
    for n in (8, 16, 32, 64):
        x = torch.rand(n, 1, 224, 224, device="cuda")
        t = timeit(lambda: ops.conv_first_fwd(x, w, torch.bfloat16, want_stats=True), 20, 3)
        y, _ = ops.conv_first_fwd(x, w, torch.bfloat16, want_stats=True)
        dy = torch.randn_like(y)
        t2 = timeit(lambda: ops.conv_first_wgrad(x, dy), 20, 3)
        print(f"  N={n:3d}: fwd {t * 1e3:6.1f} us ({y.numel() * 2 / t / 1e9:6.1f} GB/s of output)   wgrad {t2 * 1e3:6.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for mfma in ("1", "0"):
            print(f"CY_FIRST_MFMA={mfma}")
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CY_FIRST_MFMA=mfma), check=True)
