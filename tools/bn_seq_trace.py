#!/usr/bin/env python3
"""device times of the BatchNorm backward IN SEQUENCE, finalize path (reduce -> finalize -> apply) against accumulator
path (reduce_acc -> apply_fold), caches flushed before each sequence; run under rocprofv3 --kernel-trace and read the
trace with tools/bn_seq_trace.py --read <dir>"""
import sys
from pathlib import Path

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import csv
    import glob
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = []
    for r in rows:
        n = r["Kernel_Name"]
        if "ill" in n and "bn_" not in n:
            if seq:
                print("  ".join(f"{a}:{b:.1f}" for a, b in seq))
            seq = []
            continue
        short = "reduce" if "bwd_reduce" in n else ("bwdfin" if "bwd_finalize" in n else ("apply" if "bwd_apply" in n else n[:12]))
        seq.append((short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    sys.exit(0)

import torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops
dev, dt = "cuda", torch.bfloat16
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for N, C, H in ((16, 512, 14), (16, 256, 28), (16, 128, 56), (16, 64, 112), (16, 32, 224)):
    y = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
    da = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
    coef = torch.rand(5, C, device=dev) + 0.5
    for rep in range(2):
        flush.fill_(rep)
        torch.cuda.synchronize()
        ops.bn_relu_bwd(da, y, coef[0], coef[1], coef[2], coef[3], True)      # reduce, finalize, apply
        torch.cuda.synchronize()
        flush.fill_(rep + 7)
        torch.cuda.synchronize()
        ops.bn_relu_bwd_acc(da, y, coef[0], True)                             # (fill of the accumulator,) reduce_acc, apply_fold
        torch.cuda.synchronize()
