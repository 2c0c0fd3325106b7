#!/usr/bin/env python3
"""development aid: the persistent producer / consumer conv kernel (cy_conv3x3_pc_fwd) against torch CPU on
the C2 layers, and its time next to the planned kernel's:  python tools/diag_pc.py [N] [check|time|both]"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops  # noqa: E402
from tests import c2_layers as cl  # noqa: E402
from tests.test_gpu_c2_geometry import _case, _conv_input, nhwc, cpu  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
what = sys.argv[2] if len(sys.argv) > 2 else "both"
only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None
BF = torch.bfloat16
DEV = "cuda"


def err(a, b):
    a, b = cpu(a), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tot = {"plane_f": 0.0, "pc_f": 0.0, "plane_d": 0.0, "pc_d": 0.0}
for layer in cl.unet_layers(224, 512):
    name, H, C1, C2, Cout, mode, pro = layer
    if only and name not in only:
        continue
    x1, x2, w, dy, scale, shift = _case(N, layer, BF, 5)
    kw = dict(mode=mode, scale=None if scale is None else scale.to(DEV), shift=None if shift is None else shift.to(DEV))
    g1, g2 = nhwc(x1, BF), None if x2 is None else nhwc(x2, BF)
    gdy = nhwc(dy, BF)
    wf, wd = ops.pack_weights(w.to(DEV), BF)
    pf, pd = ops.pack_weights_pc(w.to(DEV), BF)
    line = f"{name:10s}"
    if what in ("check", "both"):
        ref = F.conv2d(_conv_input(x1, x2, mode, scale, shift, BF), w, None, 1, 1)
        out, stats = ops.conv3x3_pc_fwd(g1, g2, pf, Cout, **kw)
        torch.cuda.synchronize()
        o = cpu(out).double()
        s = cpu(stats).double().sum(0)
        cnt = N * H * H
        e_s1 = ((s[0] - o.sum(dim=(0, 2, 3))).abs().max() / cnt).item()
        e_s2 = ((s[1] - (o * o).sum(dim=(0, 2, 3))).abs().max() / cnt).item()
        din, _ = ops.conv3x3_pc_fwd(gdy, None, pd, C1 + C2, want_stats=False)
        e_d = err(din, F.conv_transpose2d(dy, w, None, 1, 1))
        line += f" fwd {err(out, ref):.2e} stat {e_s1:.1e} {e_s2:.1e} dgrad {e_d:.2e}"
        if C2:
            (d1, d2), _ = ops.conv3x3_pc_fwd(gdy, None, pd, C1 + C2, want_stats=False, split=C1)
            rd = F.conv_transpose2d(dy, w, None, 1, 1)
            line += f" split {err(d1, rd[:, :C1]):.2e} {err(d2, rd[:, C1:]):.2e}"
    if what in ("time", "both"):
        fl = 2.0 * N * H * H * 9 * (C1 + C2) * Cout
        t_pl = timeit(lambda: ops.conv3x3_fwd(g1, g2, wf, Cout, **kw))
        t_pc = timeit(lambda: ops.conv3x3_pc_fwd(g1, g2, pf, Cout, **kw))
        d_pl = timeit(lambda: ops.conv3x3_fwd(gdy, None, wd, C1 + C2, want_stats=False))
        d_pc = timeit(lambda: ops.conv3x3_pc_fwd(gdy, None, pd, C1 + C2, want_stats=False))
        tot["plane_f"] += t_pl; tot["pc_f"] += t_pc; tot["plane_d"] += d_pl; tot["pc_d"] += d_pc
        line += f" | fwd plane {t_pl:6.1f} pc {t_pc:6.1f} us ({fl / t_pc / 1e6:6.0f} TF) | dgrad plane {d_pl:6.1f} pc {d_pc:6.1f} us ({fl / d_pc / 1e6:6.0f} TF)"
    print(line, flush=True)
print({k: round(v, 1) for k, v in tot.items()})
