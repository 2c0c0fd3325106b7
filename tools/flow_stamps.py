#!/usr/bin/env python3
"""development aid: where a flow-kernel launch spends its time (library built with
CY_HIPCC_EXTRA_CY_CONV3X3=-DCY_FLOW_STAMPS):  python tools/flow_stamps.py <layer> <N> [fwd|dgrad]
Per workgroup: shader clocks from start to tables ready / first chunk landed / loop end / epilogue issued, and
wall-clock (100 MHz) start / end / stores acknowledged relative to the first workgroup's start."""
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
which = sys.argv[3] if len(sys.argv) > 3 else "fwd"
layer = [l for l in cl.unet_layers(224, 512) if l[0] == name][0]
_, H, C1, C2, Cout, mode, pro = layer
BF = torch.bfloat16
x1, x2, w, dy, scale, shift = _case(N, layer, BF, 5)
kw = dict(mode=mode, scale=None if scale is None else scale.cuda(), shift=None if shift is None else shift.cuda())
g1, g2 = nhwc(x1, BF), None if x2 is None else nhwc(x2, BF)
wf, wd = ops.pack_weights(w.cuda(), BF)
gdy = nhwc(dy, BF)
plan = (ops.conv3x3_plan(N, H, H, C1, C2, Cout, BF, mode, bool(pro)) if which == "fwd"
        else ops.conv3x3_plan(N, H, H, Cout, 0, C1 + C2, BF, 0, False))
print(name, N, which, plan)
nwg = plan["workgroups"]
stamps = torch.zeros(max(nwg, 1) * 8, dtype=torch.int64, device="cuda")


def run():
    if which == "fwd":
        ops.conv3x3_fwd(g1, g2, wf, Cout, **kw)
    else:
        ops.conv3x3_fwd(gdy, None, wd, C1 + C2, want_stats=False)


for _ in range(5):
    run()
torch.cuda.synchronize()
_lib.call("cy_debug_conv_stamps", stamps.data_ptr())
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
run()
e.record()
torch.cuda.synchronize()
_lib.call("cy_debug_conv_stamps", 0)
t = stamps.view(-1, 8).cpu().double()
t = t[t[:, 1] > 0]
print(f"host events: {s.elapsed_time(e) * 1e3:.1f} us for {t.shape[0]} stamped workgroups")
rt0 = t[:, 0].min()


def q(v):
    v = v.sort().values
    n = len(v)
    return f"min {v[0]:8.0f}  med {v[n // 2]:8.0f}  p90 {v[int(n * 0.9)]:8.0f}  max {v[-1]:8.0f}"


print("shader clocks since the workgroup's start:")
for k, what in ((2, "tables ready"), (3, "first chunk landed"), (4, "loop end"), (5, "epilogue issued")):
    print(f"  {what:20s} {q(t[:, k] - t[:, 1])}")
print("  loop (3 -> 4)        ", q(t[:, 4] - t[:, 3]))
print("  epilogue (4 -> 5)    ", q(t[:, 5] - t[:, 4]))
print("wall clock, us since the first workgroup's start:")
print("  start               ", q((t[:, 0] - rt0) / 100))
print("  end of code         ", q((t[:, 6] - rt0) / 100))
print("  stores acknowledged ", q((t[:, 7] - rt0) / 100))
clk = (t[:, 5] - t[:, 1]) / ((t[:, 6] - t[:, 0]) / 100)
print("  shader clock (MHz)  ", q(clk))
