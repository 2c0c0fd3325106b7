#!/usr/bin/env python3
"""How long does the host need to ISSUE one training step (no GPU sync) vs the GPU time?"""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
ctx = bench.build_step(dev, 0, 16, 16, 224, 512)
bench.run_epoch(ctx, dev, 5, 0)
torch.cuda.synchronize()
for k in (20,):
    t0 = time.perf_counter()
    bench.run_epoch(ctx, dev, k, 1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"steps={k}: host issue {1e3 * (t1 - t0) / k:.2f} ms/step, total {1e3 * (t2 - t0) / k:.2f} ms/step")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
bench.run_epoch(ctx, dev, 10, 2)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
