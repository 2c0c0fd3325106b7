#!/usr/bin/env python3
"""the dense projector over feature-map channel counts (the decoder taps of UNet(max_channel=512)), N = 32, 256 hidden units:
    python tools/bench_dense_channels.py            (on the GPU box; times are host-paced events in ms -> us)"""
import sys, torch
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd")); sys.path.insert(0, str(REPO / "tools"))
from cyhip import ops
from bench_next import timed
dev, dt = "cuda", torch.bfloat16
for (n, c, hw) in ((32, 32, 224), (32, 64, 112), (32, 128, 56), (32, 256, 28)):
    s, hid = 20 if hw >= 112 else 14, 256
    x = torch.randn(n, hw, hw, c, device=dev).to(dt).permute(0, 3, 1, 2)
    w1 = torch.randn(hid, c, device=dev) * 0.1
    b1 = torch.randn(hid, device=dev) * 0.1
    ms = timed(lambda: ops.dense_proj_fwd(x, w1, b1, (s, s), None))
    g = torch.randn(n * s * s, hid, device=dev)
    ms2 = timed(lambda: ops.dense_proj_bwd(x, w1, b1, (s, s), None, g, True, True), iters=3)
    xb = x.numel() * 2
    print(f"C={c:3d} {hw}x{hw}: fwd {ms*1e3:7.1f} us ({xb/ms/1e9:5.2f} TB/s of x)  bwd {ms2*1e3:8.1f} us ({3*xb/ms2/1e9:5.2f} TB/s)")
