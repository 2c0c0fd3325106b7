"""Every conv / weight-gradient kernel instantiation that the benchmark profile (profiles/) shows is reached
by a parity case of tests/test_gpu_c2_geometry.py: the launch plan is a pure host-side function of the layer
geometry (cy_conv3x3_plan / cy_conv3x3_wgrad_plan), so the coverage claim is checked here, on the CPU, and
the GPU test asserts per case that the plan it ran is the one listed here."""
import torch

from tests import c2_layers as cl


def _tested_kernel_names():
    from cyhip import ops
    names = set()
    for N in (16, 32):
        for name, H, C1, C2, Cout, mode, pro in cl.unet_layers(224, 512):
            names.add(cl.conv_kernel_name(ops.conv3x3_plan(N, H, H, C1, C2, Cout, torch.bfloat16, mode, pro),
                                          cin=C1 + C2, stats=True, pro=bool(pro)))
            names.add(cl.conv_kernel_name(ops.conv3x3_plan(N, H, H, Cout, 0, C1 + C2, torch.bfloat16, 0, 0), cin=Cout))
            names.add(cl.wgrad_kernel_name(ops.conv3x3_wgrad_plan(N, H, H, C1, C2, Cout, torch.bfloat16, mode, pro)))
    for name, H, C1, C2, Cout, mode, pro in cl.unet_layers(224, 512):
        if name in cl.ENCODER:  # the two passes of the two-stage step in one launch
            names.add(cl.wgrad_kernel_name(ops.conv3x3_wgrad_plan(16, H, H, C1, C2, Cout, torch.bfloat16, mode, pro,
                                                                  n_b=32)))
    return names


def test_every_profiled_conv_instantiation_has_a_parity_case():
    prof = cl.profiled_conv_kernels(cl.latest_profile())
    assert prof, "no conv kernels found in the profile summary"
    tested = _tested_kernel_names()
    missing = sorted(prof - tested)
    assert not missing, f"{cl.latest_profile().name} names kernel instantiations no C2-geometry parity case reaches: {missing}"


def test_plan_query_matches_partials_and_split():
    from cyhip import ops
    p = ops.conv3x3_plan(16, 28, 28, 256, 0, 256, torch.bfloat16, 0, 1)  # Conv4b at N=16: four-wave 16 x 64 tiles
    assert p["kernel"] == "conv3x3_flow_kernel" and p["bn"] == 64 and p["th"] == 16 and p["ksplit"] == 1
    assert p["workgroups"] == 224
    pd = ops.conv3x3_plan(16, 14, 14, 512, 0, 256, torch.bfloat16, 0, 0)  # Conv5a data gradient at N=16: 56 such
    assert pd["kernel"] == "conv3x3_flow_kernel" and pd["bn"] == 128 and pd["ksplit"] == 8  # tiles: split-K stays
    p5 = ops.conv3x3_plan(16, 14, 14, 512, 0, 512, torch.bfloat16, 0, 1)  # Conv5b at N=16: split-K over 8-chunk ranges
    assert p5["kernel"] == "conv3x3_flow_kernel" and p5["ksplit"] == 4 and p5["workgroups"] == 224
    q = ops.conv3x3_plan(16, 28, 28, 128, 0, 256, torch.bfloat16, 1, 0)  # Conv4a: 2x2 max on load stays on the plane kernel
    assert q["kernel"] == "conv3x3_plane_kernel" and q["bn"] == 128
    w = ops.conv3x3_wgrad_plan(16, 14, 14, 512, 0, 512, torch.bfloat16, 0, 1)
    assert w["splits"] >= 1 and w["workgroups"] >= 64
