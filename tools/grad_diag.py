#!/usr/bin/env python3
"""Diagnostic: per-parameter gradient error of the HIP U-Net (f32 mode) vs the f64 oracle."""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from contrastyou.arch import UNet  # noqa: E402
from cyhip.functions import SoftmaxKLFn  # noqa: E402
from oracle import losses as ol  # noqa: E402
from oracle import unet as ou  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def main(hw=224, batch=2, maxc=128):
    sd = ou.init_state_dict(1, 4, maxc, seed=21)
    g = torch.Generator().manual_seed(hw)
    x = torch.rand(batch, 1, hw, hw, generator=g)
    tgt = torch.randint(0, 4, (batch, hw, hw), generator=g)
    res = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        sdr = ou.clone_state_dict(sd, requires_grad=True, dtype=dt)
        feats = {}
        ref = ou.unet_forward(sdr, x.to(dt), training=True, momentum=0.01, feats=feats)
        for f in feats.values():
            f.retain_grad()
        ol.sup_loss(ref, tgt).backward()
        res[name] = (sdr, feats)
    net = UNet(input_dim=1, num_classes=4, max_channel=maxc, momentum=0.01)
    net.load_state_dict(sd)
    net.cuda().train()
    gfeats = {}
    def mk(n):
        def hook(m, i, o):
            o.retain_grad()
            gfeats[n] = o
        return hook

    hooks = [net.get_module(n).register_forward_hook(mk(n)) for n in net.arch_elements]
    out = net(x.cuda())
    SoftmaxKLFn.apply(out, tgt.cuda(), 1e-16).backward()
    print(f"{'param':34s} {'gpu vs f64':>11s} {'cpu32 vs f64':>12s}")
    for n, p in net.named_parameters():
        print(f"{n:34s} {rel(p.grad, res['f64'][0][n].grad):11.2e} {rel(res['f32'][0][n].grad, res['f64'][0][n].grad):12.2e}")
    print("block-output gradients:")
    for n in net.arch_elements:
        if gfeats[n].grad is not None:
            print(f"{n:12s} fwd {rel(gfeats[n], res['f64'][1][n]):9.2e}  dgrad {rel(gfeats[n].grad, res['f64'][1][n].grad):9.2e}"
                  f"   cpu32 dgrad {rel(res['f32'][1][n].grad, res['f64'][1][n].grad):9.2e}")


if __name__ == "__main__":
    main(*[int(v) for v in sys.argv[1:]])
