"""Development aid: the dense projector's matrix-core kernels (all bins = cell partition, and the same bins as a
list = per-bin jobs + colour classes) against a plain torch evaluation on the device; prints where they differ."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "contrast-you_amd"))
from cyhip import ops  # noqa: E402


def main():
    dev = "cuda"
    n, c, hw, s, hid = int(os.environ.get("N", 2)), 32, int(os.environ.get("HW", 45)), int(os.environ.get("S", 8)), 256
    dt = torch.bfloat16
    g0 = torch.Generator().manual_seed(1)
    x = torch.randn(n, hw, hw, c, generator=g0).to(dt)
    w1 = (torch.randn(hid, c, generator=g0) * 0.2).to(dt).float()
    b1 = torch.randn(hid, generator=g0) * 0.1
    dh = torch.randn(n * s * s, hid, generator=g0)
    xd = x.to(dev).permute(0, 3, 1, 2)
    w1d, b1d, dhd = w1.to(dev), b1.to(dev), dh.to(dev)

    # torch reference (f64 on the host)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w1.double().requires_grad_(True)
    br = b1.double().requires_grad_(True)
    pre = F.conv2d(xr, wr.view(hid, c, 1, 1), br)
    hp = F.adaptive_avg_pool2d(F.leaky_relu(pre, 0.01), (s, s))  # [n, hid, s, s]
    hp_rows = hp.permute(0, 2, 3, 1).reshape(n * s * s, hid)
    (hp_rows * dh.double()).sum().backward()

    def report(tag, got, ref):
        got = got.double().cpu()
        err = (got - ref).abs()
        print(f"{tag:28s} max err {err.max().item():.3e}  scale {ref.abs().max().item():.3e}")
        return err

    out = ops.dense_proj_fwd(xd, w1d, b1d, (s, s), None)
    report("fwd all bins", out, hp_rows.detach())
    allbins = np.asarray([(i, a, b) for i in range(n) for a in range(s) for b in range(s)], dtype=np.int32)
    bt = ops._bins_tensor(allbins, dev)
    out2 = ops.dense_proj_fwd(xd, w1d, b1d, (s, s), bt)
    report("fwd bin list", out2, hp_rows.detach())

    for tag, bins in (("cells", None), ("list", bt)):
        dx, dw, db = ops.dense_proj_bwd(xd, w1d, b1d, (s, s), bins, dhd, True, True)
        e = report(f"dx {tag}", dx, xr.grad)
        if e.max() > 1e-2 * xr.grad.abs().max():
            bad = (e.amax(1) > 1e-2 * xr.grad.abs().max()).nonzero()
            print("   bad pixels (n, h, w):", bad[:12].tolist(), "count", len(bad))
            hs = sorted(set(bad[:, 1].tolist()))
            ws = sorted(set(bad[:, 2].tolist()))
            print("   rows", hs[:40], "\n   cols", ws[:40])
        report(f"dw {tag}", dw, wr.grad)
        report(f"db {tag}", db, br.grad)
        print("   db got", db[:6].tolist(), "\n   db ref", br.grad[:6].tolist())
        print("   ratio", (db.double().cpu() / br.grad)[:8].tolist())
    # two neighbouring bins, one at a time and together
    def run(lst):
        b = ops._bins_tensor(np.asarray(lst, dtype=np.int32), dev)
        gsel = torch.stack([dhd[(i * s + a) * s + bb] for i, a, bb in lst])
        dx, _, _ = ops.dense_proj_bwd(xd, w1d, b1d, (s, s), b, gsel, True, False)
        return dx.float().cpu()
    A, B = run([(0, 0, 0)]), run([(0, 0, 1)])
    AB = run([(0, 0, 0), (0, 0, 1)])
    col = -((-hw) // s) - 1  # last column of bin (0, 0) = first of bin (0, 1) when they overlap
    print("shared column", col)
    print("  A ", A[0, :4, 0, col].tolist(), "\n  B ", B[0, :4, 0, col].tolist(), "\n  AB", AB[0, :4, 0, col].tolist())
    print("  AB - (A + B) max", (AB - (A + B)).abs().max().item(), " A max", A.abs().max().item())
    print("bin row edges:", [((i * hw) // s, -((-(i + 1) * hw) // s)) for i in range(s)])


if __name__ == "__main__":
    main()
