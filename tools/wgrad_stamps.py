#!/usr/bin/env python3
"""development aid (library built with CY_HIPCC_EXTRA=-DCY_WGRAD_STAMPS): in-kernel timeline of workgroup (0, 0)
of the twelve-wave weight-gradient kernel:  python tools/wgrad_stamps.py <layer> <N>
per tile and wave: request issue, MFMA loop, commit, barrier wait (shader clocks)"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import _lib, ops  # noqa: E402
from tests import c2_layers as cl  # noqa: E402
from tests.test_gpu_c2_geometry import _case, nhwc  # noqa: E402

name, N = sys.argv[1], int(sys.argv[2])
layer = [l for l in cl.unet_layers(224, 512) if l[0] == name][0]
_, H, C1, C2, Cout, mode, pro = layer
BF = torch.bfloat16
x1, x2, w, dy, scale, shift = _case(N, layer, BF, 5)
kw = dict(mode=mode, scale=None if scale is None else scale.cuda(), shift=None if shift is None else shift.cuda())
g1, g2 = nhwc(x1, BF), None if x2 is None else nhwc(x2, BF)
gdy = nhwc(dy, BF)
print(ops.conv3x3_wgrad_plan(N, H, H, C1, C2, Cout, BF, mode, bool(pro)))
stamps = torch.zeros(12 * 128, dtype=torch.int64, device="cuda")
_lib.call("cy_debug_wgrad_stamps", stamps.data_ptr())
for _ in range(3):
    ops.conv3x3_wgrad(g1, g2, gdy, **kw)
torch.cuda.synchronize()
buf = stamps.cpu().tolist()
print("per wave, cycles per tile: request, mfma loop, commit, barrier   (tiles)")
for wv in range(12):
    c = buf[wv * 8:wv * 8 + 5]
    n = max(c[4], 1)
    print(f"wave {wv:2d}: {c[0] / n:8.0f} {c[1] / n:8.0f} {c[2] / n:8.0f} {c[3] / n:8.0f}   ({c[4]})")
