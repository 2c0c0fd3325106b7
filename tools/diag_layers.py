#!/usr/bin/env python3
"""diagnostic: run every C2 layer (fwd / dgrad / wgrad parity vs torch CPU) for a dtype and batch size and
print which ones fail:  python tools/diag_layers.py f32 32"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from tests import c2_layers as cl  # noqa: E402
from tests.test_gpu_c2_geometry import _run_layer  # noqa: E402

dt = torch.float32 if sys.argv[1] == "f32" else torch.bfloat16
N = int(sys.argv[2])
hw = int(sys.argv[3]) if len(sys.argv) > 3 else 224
for layer in cl.unet_layers(hw, 512):
    try:
        plan, dplan = _run_layer(N, layer, dt, seed=5)
        print("ok  ", layer[0], plan, flush=True)
    except AssertionError as e:
        print("FAIL", layer[0], str(e)[:300], flush=True)
