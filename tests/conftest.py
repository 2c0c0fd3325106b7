import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
for p in (REPO, REPO / "contrast-you_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = REPO / "tests" / "golden"

# the fused data gradient (cy_conv3x3_dgrad_bn) is opt-in and, by its launch-plan rule, limited to the 128-cout tilings;
# the parity tests want every tiling that can take it (read once by the library, at its first plan query)
os.environ.setdefault("CY_DGRAD_BN_ALL", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
