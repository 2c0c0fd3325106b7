#!/usr/bin/env python3
"""Where the GPU's time goes inside one training step of bench.py's workload, on the GPU's own clock.

One-thread stamp kernels (cy_debug_stamp: device wall clock, 100 MHz) are enqueued at phase borders --
also inside the captured HIP graphs, where kernel tracing would serialise the branches:

    step start | labeled pass fwd (main branch) | unlabeled pass fwd (side branch) | losses + hooks
    (eager) | backward graph | optimizer

    python tools/step_timeline.py [--steps 40]        (run on the GPU box)
"""
import argparse
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
import bench  # noqa: E402
from cyhip import _lib  # noqa: E402

SLOTS = ["step_start", "A_fwd_start", "A_fwd_end", "B_fwd_start", "B_fwd_end", "pre_backward", "post_backward",
         "post_optimizer"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = bench.build_step(dev, 0, 16, 16, 224, 512)
    buf = torch.zeros(len(SLOTS), dtype=torch.int64, device=dev)
    log = []

    def stamp(name):
        _lib.call("cy_debug_stamp", buf.data_ptr(), SLOTS.index(name), torch.cuda.current_stream().cuda_stream)

    model = ctx["model"]
    orig_forward = type(model).forward
    calls = {"n": 0}

    def forward(self, x, *args, **kw):
        which = "A" if calls["n"] % 2 == 0 else "B"
        calls["n"] += 1
        stamp(f"{which}_fwd_start")
        y = orig_forward(self, x, *args, **kw)
        stamp(f"{which}_fwd_end")
        return y

    type(model).forward = forward

    from semi_seg.epochers import SemiSupervisedEpocher as Ep
    o_zero, o_scale, o_step = Ep.optimizer_zero, Ep.scale_loss, Ep.optimizer_step

    def optimizer_zero(self, *x, **k):
        log.append(buf.clone())  # device-side snapshot of the previous step's stamps: no host sync
        stamp("step_start")
        return o_zero(self, *x, **k)

    def scale_loss(self, loss):
        stamp("pre_backward")
        return o_scale(self, loss)

    def optimizer_step(self, *x, **k):
        stamp("post_backward")
        r = o_step(self, *x, **k)
        stamp("post_optimizer")
        return r

    Ep.optimizer_zero, Ep.scale_loss, Ep.optimizer_step = optimizer_zero, scale_loss, optimizer_step
    bench.run_epoch(ctx, dev, a.steps, 0)
    torch.cuda.synchronize()
    rows = torch.stack(log[a.steps // 2:]).cpu().double()  # replayed steps only
    t = {n: rows[:, i] for i, n in enumerate(SLOTS)}
    us = lambda x: (x.mean().item() / 100.0)  # noqa: E731  100 MHz ticks -> us
    fwd_begin = torch.minimum(t["A_fwd_start"], t["B_fwd_start"])
    fwd_end = torch.maximum(t["A_fwd_end"], t["B_fwd_end"])
    print(f"steps analysed: {rows.shape[0]}")
    print(f"step_start -> forward graph begins   {us(fwd_begin - t['step_start']):9.1f} us   (zero_grad, input prep)")
    print(f"labeled pass forward  (main branch)  {us(t['A_fwd_end'] - t['A_fwd_start']):9.1f} us")
    print(f"unlabeled pass forward (side branch) {us(t['B_fwd_end'] - t['B_fwd_start']):9.1f} us")
    print(f"forward graph, both branches         {us(fwd_end - fwd_begin):9.1f} us")
    print(f"losses + hooks (eager launches)      {us(t['pre_backward'] - fwd_end):9.1f} us")
    print(f"backward (graph replay)              {us(t['post_backward'] - t['pre_backward']):9.1f} us")
    print(f"optimizer                            {us(t['post_optimizer'] - t['post_backward']):9.1f} us")
    print(f"step_start -> post_optimizer         {us(t['post_optimizer'] - t['step_start']):9.1f} us")
    print(f"step_start -> next step_start        {us(t['step_start'][1:] - t['step_start'][:-1]):9.1f} us")


if __name__ == "__main__":
    main()
