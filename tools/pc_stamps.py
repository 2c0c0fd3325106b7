#!/usr/bin/env python3
"""development aid: in-kernel timeline of workgroup 0 of the producer / consumer conv kernel
(CY_PC_DEBUG=16):  python tools/pc_stamps.py <layer> <N> [fwd|dgrad]"""
import ctypes as C
import os
import sys
from pathlib import Path

import torch  # noqa: E402

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import _lib, ops  # noqa: E402
from tests import c2_layers as cl  # noqa: E402
from tests.test_gpu_c2_geometry import _case, nhwc  # noqa: E402

name, N = sys.argv[1], int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "fwd"
layer = [l for l in cl.unet_layers(224, 512) if l[0] == name][0]
_, H, C1, C2, Cout, mode, pro = layer
BF = torch.bfloat16
x1, x2, w, dy, scale, shift = _case(N, layer, BF, 5)
kw = dict(mode=mode, scale=None if scale is None else scale.cuda(), shift=None if shift is None else shift.cuda())
g1, g2 = nhwc(x1, BF), None if x2 is None else nhwc(x2, BF)
pf, pd = ops.pack_weights_pc(w.cuda(), BF)
gdy = nhwc(dy, BF)
stamps = torch.zeros(12 * 128, dtype=torch.int64, device="cuda")
_lib.call("cy_debug_pc_stamps", stamps.data_ptr())
for _ in range(3):
    if which == "fwd":
        ops.conv3x3_pc_fwd(g1, g2, pf, Cout, **kw)
    else:
        ops.conv3x3_pc_fwd(gdy, None, pd, C1 + C2, want_stats=False)
torch.cuda.synchronize()
buf = stamps.cpu().tolist()
st = [[buf[w * 128 + k] for k in range(128)] for w in range(12)]
PW = 8 if (st[8][0] or st[8][1]) else 4

print("consumer wave 0: (wait-at-barrier, stage) cycles per stage; E = epilogue marker")
c = st[0]
row = []
k = 0
while k + 2 < 128 and c[k + 1]:
    row.append((c[k + 1] - c[k], c[k + 2] - c[k + 1]))
    k += 2
    if len(row) >= 40:
        break
print(row)
print(f"producer wave {PW}: (b_issue, commit/request, barrier wait) cycles per interval")
p = st[PW]
row = []
k = 0
while k + 3 < 128 and p[k + 3]:
    row.append((p[k + 1] - p[k], p[k + 2] - p[k + 1], p[k + 3] - p[k + 2]))
    k += 3
    if len(row) >= 40:
        break
print(row)
