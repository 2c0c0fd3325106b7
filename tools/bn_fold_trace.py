#!/usr/bin/env python3
"""device time of the accumulator -> coefficients step alone (cy_bn_fold_coef: one workgroup gathers the replicas through
LDS and derives the coefficients), under rocprofv3 --kernel-trace"""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops
dev = "cuda"
for C, R in ((512, 4), (256, 4), (128, 8), (64, 16), (32, 32), (32, 1), (512, 1)):
    t = torch.zeros(R * 4 * C + R, dtype=torch.int64, device=dev)
    acc = ops.BnAccBuf(t, R, C)
    st = ops.BnState(acc, torch.ones(C, device=dev), torch.zeros(C, device=dev), 1000, 1e-5, dev)
    for _ in range(3):
        ops.bn_fold_coef(st)
        torch.cuda.synchronize()
