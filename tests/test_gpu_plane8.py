"""The experimental eight-wave plane kernel (csrc/cy_conv_plane8.h; CY_PLANE8=1, off by default): the planner
reads the switch once per process, so the C2-geometry parity cases run in a child interpreter with it set --
(a) weights through registers, via the ordinary test file, (b) weights (and, for layers without a load
transform, the halo tile) by LDS-DMA from the stage-contiguous image, via tools/p8_check.py."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]


def _run(args):
    env = dict(os.environ, CY_PLANE8="1")
    return subprocess.run([sys.executable, *args], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)


def test_plane8_register_path_c2_layers():
    r = _run(["-m", "pytest", "tests/test_gpu_c2_geometry.py", "-x", "-q", "-k",
              "c2_layer_bf16 and 16 and (Conv3b or Conv5b or Up5 or Up_conv5a or Up4 or Up_conv4b or Up3)"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_plane8_dma_path_c2_layers():
    r = _run(["tools/p8_check.py", "8"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if "conv3x3_plane8_kernel" in ln]
    assert len(lines) >= 9 and all(ln.rstrip().endswith("OK") for ln in lines), r.stdout[-3000:]
