"""Every A/B switch that selects a kernel or a launch structure, exercised once: two C2-shaped layers
(forward + statistics, data gradient, weight gradient, paired weight gradient) and a small two-pass U-Net step run in a
subprocess under the switch and are compared with the default setting's results -- the non-default branches compute the
same function, so they cannot rot unnoticed.  (The library reads its switches once per process, hence subprocesses.)"""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]

SWITCHES = [
    {"CY_FLOW": "0"}, {"CY_STREAM": "0"}, {"CY_CONV_PLANE": "0"}, {"CY_FIRST_MFMA": "0"}, {"CY_FIRST_WGRAD_MFMA": "0"}, {"CY_PLANE_XCD": "0"},
    {"CY_WGRAD_SPEC": "0"}, {"CY_WGRAD_DMA": "0"}, {"CY_WGRAD_BLK": "0"},
    {"CY_PAIR_WGRAD": "0"}, {"CY_POOL_BN_FUSE": "0"}, {"CY_BN_ACC": "0"}, {"CY_BN_FOLD_IN_KERNEL": "0"},
    {"CY_DGRAD_BN": "1"}, {"CY_DGRAD_DZ": "1"}, {"CY_ASYNC_WGRAD": "0"}, {"CY_TWO_STREAM": "0"},
]


def _run(tmp_path, env_extra, name):
    out = tmp_path / f"{name}.pt"
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("CY_") and k not in ("CY_DGRAD_BN_ALL",):
            del env[k]
    env.update(env_extra)
    r = subprocess.run([sys.executable, str(REPO / "tests" / "switch_case.py"), str(out)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(out, weights_only=True)


@pytest.fixture(scope="module")
def default_result(tmp_path_factory):
    return _run(tmp_path_factory.mktemp("switches"), {}, "default")


@pytest.mark.parametrize("sw", SWITCHES, ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()))
def test_switch_computes_the_same_function(sw, default_result, tmp_path):
    res = _run(tmp_path, sw, "case")
    assert res.keys() == default_result.keys()
    for k, ref in default_result.items():
        if "CY_FIRST_MFMA" in sw and k.startswith("unet_"):
            # the VALU first layer keeps the image and its weights in f32, the matrix-core one rounds them to the storage
            # type like autocast does: a different (documented) rounding of the INPUT, which a randomly initialised
            # 22-layer network on three slices amplifies -- the layer itself is compared above (first_fwd, first_stats)
            continue
        got = res[k]
        scale = ref.abs().max().item() + 1e-12
        err = (got - ref).abs().max().item()
        # same inputs, same arithmetic up to accumulation order and one 16-bit rounding of an intermediate
        if k.startswith("unet_grad"):
            tol = 0.15
        elif k.startswith("unet_buf"):
            tol = 1e-4
        elif "wgrad" in k or "stats" in k:
            tol = 5e-3
        else:
            tol = 2e-2
        assert err <= tol * scale, f"{k}: {err:.3e} > {tol} * {scale:.3e} under {sw}"
