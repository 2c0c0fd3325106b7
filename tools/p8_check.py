#!/usr/bin/env python3
"""development aid: eight-wave plane kernel with LDS-DMA weights (cy_debug_p8_weights) against the default
plane kernel on C2 layers: python tools/p8_check.py [N]"""
import os
import sys
from pathlib import Path

os.environ["CY_PLANE8"] = "1"
import torch  # noqa: E402

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import _lib, ops  # noqa: E402
from tests import c2_layers as cl  # noqa: E402
from tests.test_gpu_c2_geometry import _case, _conv_input, nhwc  # noqa: E402
import torch.nn.functional as F  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
BF = torch.bfloat16
for layer in cl.unet_layers(224, 512):
    name, H, C1, C2, Cout, mode, pro = layer
    if Cout < 128 or mode == 1:
        continue
    x1, x2, w, dy, scale, shift = _case(N, layer, BF, 11)
    ref = F.conv2d(_conv_input(x1, x2, mode, scale, shift, BF), w, None, 1, 1)
    kw = dict(mode=mode, scale=None if scale is None else scale.cuda(), shift=None if shift is None else shift.cuda())
    wf, wd = ops.pack_weights(w.cuda(), BF)
    pf, pd = ops.pack_weights_pc(w.cuda(), BF)
    g1, g2 = nhwc(x1, BF), None if x2 is None else nhwc(x2, BF)
    plan = ops.conv3x3_plan(N, H, H, C1, C2, Cout, BF, mode, bool(pro))
    _lib.call("cy_debug_p8_weights", pf.data_ptr())
    out, stats = ops.conv3x3_fwd(g1, g2, wf, Cout, **kw)
    torch.cuda.synchronize()
    _lib.call("cy_debug_p8_weights", 0)
    err = (out.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    o = out.float().cpu().double()
    s = stats.float().cpu().double().sum(0)
    serr = (s[0] - o.sum(dim=(0, 2, 3))).abs().max().item() / (o.sum(dim=(0, 2, 3)).abs().max().item() + 1e-30)
    rd = F.conv_transpose2d(dy, w, None, 1, 1)
    _lib.call("cy_debug_p8_weights", pd.data_ptr())
    din, _ = ops.conv3x3_fwd(nhwc(dy, BF), None, wd, C1 + C2, want_stats=False)
    torch.cuda.synchronize()
    _lib.call("cy_debug_p8_weights", 0)
    derr = (din.float().cpu() - rd).abs().max().item() / rd.abs().max().item()
    ok = err < 1.2e-2 and serr < 1e-4 and derr < 1.2e-2
    print(f"{name:10s} {plan['kernel']} ksplit {plan['ksplit']} rel err {err:.2e} stat err {serr:.1e} dgrad err {derr:.2e}",
          "OK" if ok else "FAIL")
