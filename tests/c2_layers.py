"""The 3x3 conv layers of the reference U-Net (contrastyou/arch/unet.py:72-103) at a BASELINE geometry,
as (name, H, C1, C2, Cout, mode, prologue) tuples in forward order, and the kernel-instantiation names
the library's launch plans stand for (as they appear in profiles/*kernel_stats*).  Shared by
tests/test_plan_coverage.py (CPU: every profiled instantiation is reached by a parity case) and
tests/test_gpu_c2_geometry.py (GPU: those parity cases)."""
from __future__ import annotations

import re
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def unet_layers(hw: int = 224, max_channel: int = 512):
    c = [max_channel // 16 * m for m in (1, 2, 4, 8, 16)]
    s = [hw // (2 ** i) for i in range(5)]
    layers = [("Conv1b", s[0], c[0], 0, c[0], 0, 1)]
    for i in range(1, 5):
        layers.append((f"Conv{i + 1}a", s[i], c[i - 1], 0, c[i], 0, 0))   # reads the pooled tensor the previous block wrote
        layers.append((f"Conv{i + 1}b", s[i], c[i], 0, c[i], 0, 1))       # BN+ReLU prologue
    for i in range(3, -1, -1):
        layers.append((f"Up{i + 2}", s[i], c[i + 1], 0, c[i], 2, 0))      # nearest x2 on load
        layers.append((f"Up_conv{i + 2}a", s[i], c[i], c[i], c[i], 0, 0))  # concat (skip, up)
        layers.append((f"Up_conv{i + 2}b", s[i], c[i], 0, c[i], 0, 1))
    return layers


ENCODER = ("Conv1b", "Conv2a", "Conv2b", "Conv3a", "Conv3b", "Conv4a", "Conv4b", "Conv5a", "Conv5b")


def conv_kernel_name(plan: dict, dtype_tag: str = "DF16b", cin: int = 0, stats: bool = False, pro: bool = False) -> str:
    """mangled-name fragment of the instantiation a cy_conv_plan stands for (cin / stats / pro: what the streaming
    kernel is additionally instantiated on)"""
    b = lambda v: f"Lb{int(bool(v))}E"  # noqa: E731
    i = lambda v: f"Li{int(v)}E"  # noqa: E731
    if plan["kernel"] == "conv3x3_plane_kernel":
        wgm, wgn, pitch, allt = (2, 2, 128, 0) if plan["bn"] == 128 else (4, 1, 64, 1)
        return ("conv3x3_plane_kernelI" + dtype_tag + i(plan["th"]) + i(plan["bn"]) + i(wgm) + i(wgn) + i(pitch)
                + b(allt) + b(plan["one_per_cu"]) + "E")
    if plan["kernel"] == "conv3x3_flow_kernel":
        wgm, wgn = (4, 2) if plan["bn"] == 128 else ((4, 1) if plan["th"] in (16, 32) else (8, 1))
        return ("conv3x3_flow_kernelI" + dtype_tag + i(plan["th"]) + i(plan["bn"]) + i(wgm) + i(wgn) + b(plan["tw"] == 16) + i(0) + "E")  # (MODE 0: the forward / plain data-gradient build)
    if plan["kernel"] == "conv3x3_stream_kernel":
        return ("conv3x3_stream_kernelI" + dtype_tag + i(cin // 32) + i(plan["bn"] // 32) + b(stats) + b(pro) + "E")
    wgm, wgn = ((1, 4) if (plan["th"], plan["tw"]) in ((8, 28), (16, 14)) else (2, 2)) if plan["bn"] == 128 else (4, 1)
    pitch, allt = (128, 0) if plan["bn"] == 128 else (64, 1)
    return ("conv3x3_igemm_kernelI" + dtype_tag + "S0_" + i(plan["th"]) + i(plan["tw"]) + i(plan["bn"]) + i(wgm)
            + i(wgn) + i(pitch) + b(allt) + "E")


def wgrad_kernel_name(plan: dict) -> str:
    if plan["twelve"] == 2:
        return "wgrad12s_kernel"
    base = "wgrad12_kernel" if plan["twelve"] else "wgrad_kernel"
    return f"{base}<{plan['wco']}, {plan['wci']}, {plan['wk']}"  # (prefix: a trailing type argument may follow)


def profiled_conv_kernels(path: Path):
    """conv / weight-gradient kernel instantiations named in a `rocprofv3 --kernel-trace --stats` summary
    (profiles/rNN_bench_c2_kernel_stats_single_stream.txt).  Every row that names a conv3x3 / wgrad main kernel
    must parse into an instantiation -- a row this function cannot read would otherwise be invisible to the
    coverage gate (rocprofv3's own demangling mutilated the streaming kernel's template
    arguments; tools/profile_round.sh now keeps the names mangled)."""
    names = set()
    for line in path.read_text().splitlines():
        tok = line.split("  ")[0].strip()
        if not tok or tok.startswith("#") or tok == "kernel":
            continue
        is_conv = "conv3x3_" in tok and not any(k in tok for k in ("conv3x3_first", "splitk"))
        is_wgrad = "wgrad" in tok and not any(k in tok for k in ("reduce", "first_wgrad"))
        if not (is_conv or is_wgrad):
            continue
        found = False
        m = re.search(r"(conv3x3_(?:plane|igemm|stream|flow)_kernelI\w+?)Ev", tok)
        if m:
            names.add(m.group(1))
            found = True
        if "wgrad12s_kernel" in tok:
            names.add("wgrad12s_kernel")
            found = True
        m = re.match(r"(wgrad(?:12)?_kernel<)(?:[A-Za-z_]\w*, )?(\d+, \d+, \d+)", tok)
        if m:
            names.add(m.group(1) + m.group(2))
            found = True
        m = re.search(r"(wgrad(?:12)?_kernel)I(?:DF16[b_]|f)?Li(\d+)ELi(\d+)ELi(\d+)E", tok)  # mangled form
        if m:
            names.add(f"{m.group(1)}<{m.group(2)}, {m.group(3)}, {m.group(4)}")
            found = True
        if not found:
            raise AssertionError(f"{path.name}: cannot tell which instantiation this row is: {tok!r}")
    return names


def latest_profile() -> Path:
    files = sorted((REPO / "profiles").glob("r*_bench_c2_kernel_stats_single_stream.txt"))
    return files[-1]
