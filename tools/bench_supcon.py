#!/usr/bin/env python3
"""SupCon at the C5 size (4096 embeddings, D = 256): kernels that materialise S against the fused ones
(device time per call from a HIP graph of 10 calls):  python tools/bench_supcon.py [n]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "contrast-you_amd"))
from cyhip import ops  # noqa: E402


def timed(fn, reps=10, rounds=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(rounds):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
D = 256
z = F.normalize(torch.randn(2 * n, D, device="cuda"), dim=1)
lab = (torch.arange(n, device="cuda") % 3).int()
gs = torch.ones(1, device="cuda")
_, S, st = ops.supcon_fwd(z, lab, None, 0.07)
_, _, stf = ops.supcon_fwd_fused(z, lab, None, 0.07)
flops = 2.0 * (2 * n) ** 2 * D
for name, f, k in (("fwd  materialised", lambda: ops.supcon_fwd(z, lab, None, 0.07), 1),
                   ("fwd  fused       ", lambda: ops.supcon_fwd_fused(z, lab, None, 0.07), 1),
                   ("bwd  materialised", lambda: ops.supcon_bwd(z, lab, None, S, st, gs, 0.07), 1),
                   ("bwd  fused       ", lambda: ops.supcon_bwd_fused(z, lab, None, stf, gs, 0.07), 2)):
    t = timed(f)
    print(f"{name} {t:8.1f} us   {k * flops / t / 1e6:6.1f} TFLOP/s (f32 MFMA)")
