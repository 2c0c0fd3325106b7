#!/usr/bin/env python3
"""Kernel timings of the section-8 "next" rows at config-C2 shapes (Up_conv2 features of 2*16 slices:
[32, 32, 224, 224]); prints one line per op with achieved GB/s against the algorithmic bytes."""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def line(name, ms, nbytes=None, flops=None):
    extra = ""
    if nbytes:
        extra += f"  {nbytes / ms / 1e6:8.1f} GB/s"
    if flops:
        extra += f"  {flops / ms / 1e9:8.2f} TFLOP/s"
    print(f"{name:44s} {ms * 1e3:9.1f} us{extra}")


def main():
    dev = "cuda"
    dt = torch.bfloat16
    n, c, hw, s, hid = 32, 32, 224, 20, 256
    x = torch.randn(n, hw, hw, c, device=dev).to(dt).permute(0, 3, 1, 2)
    w1 = torch.randn(hid, c, device=dev) * 0.1
    b1 = torch.randn(hid, device=dev) * 0.1
    npix = n * hw * hw
    xbytes = npix * c * 2
    ms = timed(lambda: ops.dense_proj_fwd(x, w1, b1, (s, s), None))
    line("dense_proj_fwd all bins (12800)", ms, xbytes, 2.0 * npix * c * hid)
    g = torch.randn(n * s * s, hid, device=dev)
    ms = timed(lambda: ops.dense_proj_bwd(x, w1, b1, (s, s), None, g, True, True), iters=3)
    line("dense_proj_bwd all bins", ms, 3 * xbytes, 6.0 * npix * c * hid)
    import numpy as np
    rs = np.random.RandomState(0)
    pts = [(i, int(a), int(b)) for i in range(n) for a, b in zip(rs.choice(s, 5, replace=False), rs.choice(s, 5, replace=False))]
    bins = ops._bins_tensor(np.asarray(pts, dtype=np.int32), dev)
    ms = timed(lambda: ops.dense_proj_fwd(x, w1, b1, (s, s), bins))
    line("dense_proj_fwd 160 sampled bins", ms)
    g2 = torch.randn(len(pts), hid, device=dev)
    ms = timed(lambda: ops.dense_proj_bwd(x, w1, b1, (s, s), bins, g2, True, True))
    line("dense_proj_bwd 160 sampled bins (+zero dx)", ms)

    # cluster head 5 x 20 on the same features
    S, k = 5, 20
    w = torch.randn(S * k, c, 1, 1, device=dev) * 0.1
    b = torch.randn(S * k, device=dev) * 0.1
    ms = timed(lambda: ops.head_fwd(x, w, b))
    line("head1x1_fwd K=100", ms, xbytes + npix * S * k * 4, 2.0 * npix * c * S * k)
    logits = ops.head_fwd(x, w, b).permute(0, 2, 3, 1).reshape(npix, S * k)
    ms = timed(lambda: ops.group_softmax_fwd(logits, S, k))
    line("group_softmax_fwd", ms, 2 * npix * S * k * 4)
    probs = ops.group_softmax_fwd(logits, S, k)
    dp = torch.randn_like(probs)
    ms = timed(lambda: ops.group_softmax_bwd(probs, dp))
    line("group_softmax_bwd", ms, 3 * npix * S * k * 4)
    dl = ops.group_softmax_bwd(probs, dp).view(n, hw, hw, S * k).permute(0, 3, 1, 2)
    ms = timed(lambda: ops.head_bwd(x, w, dl, True, True), iters=3)
    line("head1x1_bwd K=100", ms, 2 * xbytes + npix * S * k * 4)
    ms = timed(lambda: ops.head_bwd(x, w, dl, True, False), iters=3)
    line("  dx only", ms)
    ms = timed(lambda: ops.head_bwd(x, w, dl, False, True), iters=3)
    line("  dw only", ms)
    rows = x.permute(0, 2, 3, 1).reshape(npix, c)
    w2 = w.reshape(S * k, c).contiguous()
    ms = timed(lambda: ops.cluster_head_fwd(rows, w2, b, S, k))
    line("cluster_head_fwd (conv1x1 + softmax, one pass)", ms, xbytes + npix * S * k * 4, 2.0 * npix * c * S * k)
    ms = timed(lambda: ops.cluster_head_bwd(rows, w2, probs, dp, 1.0, True, True))
    line("cluster_head_bwd (softmax bwd + dx + dw, one pass)", ms, 2 * xbytes + 2 * npix * S * k * 4, 6.0 * npix * c * S * k)
    half = n // 2
    a1 = probs[0, : half * hw * hw].view(half, hw, hw, k)
    a2 = probs[0, half * hw * hw:].view(half, hw, hw, k)
    ms = timed(lambda: ops.joint_fwd(a1, a2, half, hw, hw, k, 0, True))
    line("joint_fwd k=20 (16 slices, one sub-head)", ms, 2 * half * hw * hw * k * 4)
    J = ops.joint_fwd(a1, a2, half, hw, hw, k, 0, True)
    ms = timed(lambda: ops.iid_loss(J, 0, False, 1.0, 1e-5))
    line("iid_loss (single block)", ms)
    _, _, dJ = ops.iid_loss(J, 0, False, 1.0, 1e-5)
    gs = torch.ones(1, device=dev)
    ms = timed(lambda: ops.joint_bwd(a1, a2, dJ, gs, half, hw, hw, k, 0, True, True, True))
    line("joint_bwd (both inputs)", ms, 4 * half * hw * hw * k * 4)
    ms = timed(lambda: ops.joint_fwd(a1, a2, half, hw, hw, k, 1, False), iters=3)
    line("joint_fwd k=20 padding 1 (9 displacements)", ms, 2 * half * hw * hw * k * 4)

    # GroupNorm + SiLU at a UNet2-like level: 16 x 64 x 112 x 112
    N, C, H = 16, 64, 112
    y = torch.randn(N, H, H, C, device=dev).to(dt).permute(0, 3, 1, 2)
    bias, gam, bet = (torch.randn(C, device=dev) for _ in range(3))
    ms = timed(lambda: ops.gn_silu_fwd(y, bias, gam, bet, 8, 1e-5))
    line("gn_silu_fwd 16x64x112x112", ms, 3 * y.numel() * 2)
    out, mr = ops.gn_silu_fwd(y, bias, gam, bet, 8, 1e-5)
    dz = torch.randn_like(y)
    ms = timed(lambda: ops.gn_silu_bwd(y, dz, bias, gam, bet, mr, 8))
    line("gn_silu_bwd", ms, 5 * y.numel() * 2)
    img = torch.rand(32, 1, 224, 224, device=dev)
    ms = timed(lambda: ops.bilinear_fwd(img, (28, 28)))
    line("bilinear 224->28", ms)


if __name__ == "__main__":
    main()
