#!/usr/bin/env python3
"""development aid: in-kernel timeline of workgroup 0 of the streaming conv kernel (CY_STREAM=1):
    python tools/stream_stamps.py <layer> <N> [fwd|dgrad]
per tile and wave: wait for the tile's DMA, barrier, DMA issue of tile+PD, MFMAs, epilogue (shader clocks)"""
import os
import sys
from pathlib import Path

os.environ.setdefault("CY_STREAM", "1")
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
wf, wd = ops.pack_weights(w.cuda(), BF)
gdy = nhwc(dy, BF)
print(ops.conv3x3_plan(N, H, H, C1, C2, Cout, BF, mode, bool(pro)) if which == "fwd"
      else ops.conv3x3_plan(N, H, H, Cout, 0, C1 + C2, BF, 0, False))
stamps = torch.zeros(12 * 128, dtype=torch.int64, device="cuda")
_lib.call("cy_debug_conv_stamps", stamps.data_ptr())
for _ in range(3):
    if which == "fwd":
        ops.conv3x3_fwd(g1, g2, wf, Cout, **kw)
    else:
        ops.conv3x3_fwd(gdy, None, wd, C1 + C2, want_stats=False)
torch.cuda.synchronize()
buf = stamps.cpu().tolist()
for wv in (0, 1, 4):
    c = buf[wv * 128:wv * 128 + 96]
    if not c[1]:
        continue
    rows = []
    for k in range(0, 90, 6):
        if not c[k + 6]:
            break
        rows.append(tuple(c[k + j + 1] - c[k + j] for j in range(6)))
    print(f"wave {wv}: per tile (dma wait, barrier, dma issue, mfma, epilogue, loop); first stamp {c[0] - buf[0]}")
    print(rows)
