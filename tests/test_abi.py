"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/contrastyou_hip.h declares, and the product refuses to run without it / without a GPU.
No compute call is made here."""
import ctypes
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
HEADER = REPO / "include" / "contrastyou_hip.h"


def header_symbols():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cy_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cyhip import _lib
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    # and the binding declares a signature for each of them (no untyped calls)
    assert set(names) == set(_lib.exported_names())
    assert lib.cy_abi_version() == _lib.ABI_VERSION
    assert lib.cy_build_arch() == b"gfx950"


def test_conv_desc_layout_matches_header():
    from cyhip._lib import ConvDesc
    text = HEADER.read_text()
    body = re.search(r"typedef struct cy_conv_desc \{(.*?)\} cy_conv_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl.startswith("int32_t"):
            fields += [f.strip() for f in decl[len("int32_t"):].split(",")]
    assert fields == [f[0] for f in ConvDesc._fields_]
    assert ctypes.sizeof(ConvDesc) == 4 * len(fields)


def test_argument_errors_are_reported_not_launched():
    from cyhip import _lib
    lib = _lib.load()
    # NULL pointers -> CY_ERR_ARG before any launch (safe without a GPU)
    assert lib.cy_conv3x3_pack_weights(None, None, None, 8, 8, 0, None) == -1
    assert lib.cy_bn_relu_apply(None, None, None, None, 10, 8, 0, 0, None) == -1
    with pytest.raises(_lib.HipKernelError):
        _lib.call("cy_sgemm", None, None, None, 4, 4, 4, 1.0, 0, None)
    a, b = ctypes.c_int(), ctypes.c_int()
    assert lib.cy_conv3x3_packed_dims(32, 1, ctypes.byref(a), ctypes.byref(b)) == 0
    assert (a.value, b.value) == (128, 64)


def test_product_refuses_cpu_tensors():
    import torch
    from contrastyou.arch import UNet
    net = UNet(input_dim=1, num_classes=4, max_channel=128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 1, 32, 32))


def test_no_oracle_import_in_product():
    """the product tree must never import the oracle (it is test infrastructure)"""
    for p in (REPO / "contrast-you_amd").rglob("*.py"):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), p
