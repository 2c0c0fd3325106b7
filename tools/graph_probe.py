#!/usr/bin/env python3
"""Feasibility probe: capture the two network passes of a step (forward + backward, with the
side-stream forks inside) in HIP graphs via torch.cuda.make_graphed_callables and compare with eager."""
import sys
import time
from pathlib import Path

import torch
from torch import nn

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from contrastyou.arch import UNet  # noqa: E402
from cyhip import functions as F  # noqa: E402
from cyhip import ops  # noqa: E402


class TwoPass(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, xa, xb):
        dev = xa.device
        main = torch.cuda.current_stream(dev)
        side = ops.side_stream(dev, "pass2")
        side.wait_stream(main)
        ops.note_side_work(side)
        feats = {}
        h = self.model.get_module("Conv5").register_forward_hook(lambda m, i, o: feats.setdefault(len(feats), o))
        ya = self.model(xa)
        with torch.cuda.stream(side):
            yb = self.model(xb)
        main.wait_stream(side)
        h.remove()
        return ya, yb, feats[1]


def main():
    dev = "cuda"
    torch.manual_seed(0)
    model = UNet(input_dim=1, num_classes=4, max_channel=512, momentum=0.01).to(dev)
    from contrastyou.optim import RAdam
    opt = RAdam(model.parameters(), lr=1e-4)
    opt.zero_grad()
    xa = torch.rand(16, 1, 224, 224, device=dev)
    xb = torch.rand(32, 1, 224, 224, device=dev)
    tp = TwoPass(model)

    def step(mod):
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            ya, yb, f = mod(xa, xb)
            loss = ya.float().square().mean() + yb.float().mean() * 0.1 + f.float().square().mean()
        loss.backward()
        return loss

    for _ in range(3):
        opt.zero_grad()
        step(tp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        opt.zero_grad()
        step(tp)
    torch.cuda.synchronize()
    print(f"eager: {(time.perf_counter() - t0) * 100:.2f} ms / fwd+bwd")
    opt.zero_grad()
    l0 = step(tp)
    torch.cuda.synchronize()
    g_eager = opt._flat[0].grad.clone()
    bufs = {k: v.clone() for k, v in model.state_dict().items()}

    class Serial(nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model

        def forward(self, xa, xb):
            feats = []
            h = self.model.get_module("Conv5").register_forward_hook(lambda m, i, o: feats.append(o))
            ya, yb = self.model(xa), self.model(xb)
            h.remove()
            return ya, yb, feats[1]

    def report(tag, ga, gb):
        off, worst = 0, []
        for (n, p) in model.named_parameters():
            k = p.numel()
            a, b = ga[off:off + k], gb[off:off + k]
            worst.append((float((a - b).abs().max() / (b.abs().max() + 1e-30)), n))
            off += k
        worst.sort(reverse=True)
        print(tag, worst[:4])

    opt.zero_grad()
    step(Serial(model))
    torch.cuda.synchronize()
    g_serial = opt._flat[0].grad.clone()
    report("two-stream eager vs serial", g_eager, g_serial)
    opt.zero_grad()
    step(Serial(model))
    torch.cuda.synchronize()
    report("serial vs serial", opt._flat[0].grad.clone(), g_serial)
    model.load_state_dict(bufs)

    ops._order_events.clear()
    F.bump_weights_epoch()
    ops.CAPTURING = True
    with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
        gtp = torch.cuda.make_graphed_callables(tp, (xa, xb), allow_unused_input=True)
    ops.CAPTURING = False
    ops._order_events.clear()
    F.bump_weights_epoch()
    model.load_state_dict(bufs)
    torch.cuda.synchronize()
    print("graphs built")
    opt.zero_grad()
    l1 = step(gtp)
    torch.cuda.synchronize()
    g_graph = opt._flat[0].grad.clone()
    print("loss eager/graph", float(l0), float(l1))
    print("grad rel diff", float((g_eager - g_graph).abs().max() / g_eager.abs().max()))
    report("graph vs serial", g_graph, g_serial)
    for _ in range(3):
        opt.zero_grad()
        step(gtp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        opt.zero_grad()
        step(gtp)
    torch.cuda.synchronize()
    print(f"graphed: {(time.perf_counter() - t0) * 50:.2f} ms / fwd+bwd")


if __name__ == "__main__":
    main()
