#!/usr/bin/env python3
"""What a HIP event pair around ONE launch adds to the launch's duration (bench.py's instrumented pass vs rocprofv3):
per layer, (A) 20 back-to-back launches inside one event pair, (B) an event pair around each of 20 launches queued
behind a spin kernel (the GPU never waits for the host), (C) the same without the spin kernel.
Run on the GPU box:  python tools/event_overhead.py"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import _lib, ops  # noqa: E402


def main():
    dt, dev = torch.bfloat16, "cuda"
    cases = [("Up4 N=32 (flow 32x128)", 32, 56, 256, 128), ("Conv4b N=16 (flow 16x128)", 16, 28, 256, 256),
             ("Conv2b N=16 dgrad (plane/flow64)", 16, 112, 64, 64), ("Conv1b N=16 dgrad (stream)", 16, 224, 32, 32)]
    for name, N, H, Cin, Cout in cases:
        x = ops.empty_nhwc(N, Cin, H, H, dt, dev).normal_()
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        wf, _ = ops.pack_weights(w, dt)
        fn = lambda: ops.conv3x3_fwd(x, None, wf, Cout, want_stats=False)  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        K = 20
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(K):
            fn()
        e.record()
        torch.cuda.synchronize()
        a = s.elapsed_time(e) / K * 1e3
        res = {}
        for spin in (True, False):
            ops.PROFILE = []
            if spin:
                _lib.call("cy_debug_spin", 3000, ops._stream())
            for _ in range(K):
                fn()
            torch.cuda.synchronize()
            rec, ops.PROFILE = ops.PROFILE, None
            per = [r[2].elapsed_time(r[3]) * 1e3 for r in rec]
            span = rec[0][2].elapsed_time(rec[-1][3]) * 1e3 / K
            res[spin] = (sum(per) / K, min(per), max(per), span)
        print(f"{name:36s} back-to-back {a:6.1f} us | pairs behind a spin: mean {res[True][0]:6.1f} min {res[True][1]:6.1f} "
              f"max {res[True][2]:6.1f} span/launch {res[True][3]:6.1f} | pairs, no spin: mean {res[False][0]:6.1f} "
              f"span/launch {res[False][3]:6.1f}")


if __name__ == "__main__":
    main()
