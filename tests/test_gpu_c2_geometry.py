"""Parity of the conv kernels AT THE BENCHMARKED GEOMETRY: every 3x3 layer of UNet(max_channel=512) on
224 x 224 inputs (contrastyou/arch/unet.py:72-103) at the two batch sizes of the C2 step (N=16 labeled pass,
N=32 unlabeled pass), bf16 -- forward + BN statistics, data gradient, weight gradient, and the paired
(16 + 32 images) weight gradient of the encoder -- against torch's CPU convolution on the same
bf16-representable inputs.  Each case asserts the launch plan it ran, and tests/test_plan_coverage.py checks
(on the CPU) that the plans of these cases cover every instantiation named in profiles/.
Also: the C4 geometry (256 x 256, the igemm kernel: widths that are not multiples of 14) and one full-size
f32 C2 step against the oracle.

Tolerance: bf16 storage, f32 accumulation: 1.2e-2 of the tensor's max magnitude for activations
(one bf16 rounding of the output + accumulation-order noise), 2e-3 for f32 weight gradients."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests import c2_layers as cl

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def nhwc(t, dtype=None):
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def cpu(t):
    return t.detach().float().cpu()


def assert_close(a, b, rel, what):
    a, b = cpu(a), b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} > {rel:.1e} * {scale:.3e}"


def _case(N, layer, dtype, seed):
    name, H, C1, C2, Cout, mode, pro = layer
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1)  # noqa: E731
    sh = 2 * H if mode == 1 else (H // 2 if mode == 2 else H)
    x1 = r(N, C1, sh, sh).to(dtype).float()
    x2 = r(N, C2, H, H).to(dtype).float() if C2 else None
    w = (r(Cout, C1 + C2, 3, 3) / math.sqrt(9 * (C1 + C2))).to(dtype).float()
    dy = r(N, Cout, H, H).to(dtype).float()
    scale = shift = None
    if pro:
        scale, shift = r(C1) + 0.2, r(C1) * 0.3
    return x1, x2, w, dy, scale, shift


def _conv_input(x1, x2, mode, scale, shift, dtype):
    a = x1
    if scale is not None:  # the kernel rounds relu(scale*x+shift) to the storage type before the MFMA
        a = F.relu(a * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dtype).float()
    if mode == 1:
        a = F.max_pool2d(a, 2, 2)
    elif mode == 2:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    if x2 is not None:
        a = torch.cat((a, x2), 1)
    return a


def _run_layer(N, layer, dtype, seed, expect_kernel=None):
    from cyhip import ops
    name, H, C1, C2, Cout, mode, pro = layer
    x1, x2, w, dy, scale, shift = _case(N, layer, dtype, seed)
    a_in = _conv_input(x1, x2, mode, scale, shift, dtype)
    wv = w.clone().requires_grad_(True)
    ref = F.conv2d(a_in, wv, None, 1, 1)
    (ref * dy).sum().backward()
    kw = dict(mode=mode, scale=None if scale is None else scale.to(DEV), shift=None if shift is None else shift.to(DEV))
    wf, wd = ops.pack_weights(w.to(DEV), dtype)
    g1, g2 = nhwc(x1, dtype), None if x2 is None else nhwc(x2, dtype)
    plan = ops.conv3x3_plan(N, H, H, C1, C2, Cout, dtype, mode, bool(pro))
    if expect_kernel is not None:
        assert plan["kernel"] in expect_kernel, (name, plan)
    out, stats = ops.conv3x3_fwd(g1, g2, wf, Cout, **kw)
    tol = {BF: 1.2e-2, torch.float16: 2e-3}.get(dtype, 2e-5)
    assert_close(out, ref, tol, f"{name} N={N} fwd {plan}")
    assert stats.shape[0] == plan["partials"], (name, stats.shape, plan)
    o = cpu(out).double()
    s = cpu(stats).double().sum(0)
    cnt = N * H * H
    assert_close(s[0] / cnt, o.sum(dim=(0, 2, 3)) / cnt, 1e-4, f"{name} N={N} stat sum")
    assert_close(s[1] / cnt, (o * o).sum(dim=(0, 2, 3)) / cnt, 1e-4, f"{name} N={N} stat sumsq")
    dplan = ops.conv3x3_plan(N, H, H, Cout, 0, C1 + C2, dtype, 0, False)
    din, _ = ops.conv3x3_fwd(nhwc(dy, dtype), None, wd, C1 + C2, want_stats=False)
    assert_close(din, F.conv_transpose2d(dy, w, None, 1, 1), tol, f"{name} N={N} dgrad {dplan}")
    dw = ops.conv3x3_wgrad(g1, g2, nhwc(dy, dtype), **kw)
    assert_close(dw, wv.grad, 2e-3 if dtype in (BF, torch.float16) else 2e-5, f"{name} N={N} wgrad")
    return plan, dplan


C2_LAYERS = cl.unet_layers(224, 512)


@pytest.mark.parametrize("N", [16, 32])
@pytest.mark.parametrize("layer", C2_LAYERS, ids=[l[0] for l in C2_LAYERS])
def test_c2_layer_bf16(layer, N):
    plan, dplan = _run_layer(N, layer, BF, seed=1000 + N, expect_kernel=("conv3x3_plane_kernel", "conv3x3_flow_kernel", "conv3x3_stream_kernel"))
    # the claims of VERDICT r01 / ADVICE r01: these geometries reach split-K above 1, 512-channel operands and
    # (where the plane kernel is planned) the one-workgroup-per-CU build
    name = layer[0]
    if name == "Conv5b" and N == 16:
        assert plan["ksplit"] > 1, plan


STREAM_LAYERS = [l for l in C2_LAYERS if l[0] in ("Conv1b", "Up2", "Up_conv2a", "Up_conv2b")]


@pytest.mark.parametrize("dtype", [BF, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("layer", STREAM_LAYERS, ids=lambda l: l[0])
def test_c2_low_channel_layers_take_the_streaming_kernel(layer, dtype):
    """the 224 x 224 level (Cin, Cout in {32, 64}) runs in the persistent LDS-DMA kernel (csrc/cy_conv_stream.h):
    forward with BN statistics (upsample / concat / BN+ReLU-prologue loads) and data gradient (incl. the split
    epilogue of the concat layer), both 16-bit storage types"""
    plan, dplan = _run_layer(16, layer, dtype, seed=555, expect_kernel=("conv3x3_stream_kernel",))
    assert dplan["kernel"] == "conv3x3_stream_kernel", dplan
    name, H, C1, C2, Cout, mode, pro = layer
    if C2:  # the data gradient of a concat layer leaves as two tensors (skip branch / upsampled branch)
        from cyhip import ops
        x1, x2, w, dy, scale, shift = _case(16, layer, dtype, 555)
        _, wd = ops.pack_weights(w.to(DEV), dtype)
        (d1, d2), _ = ops.conv3x3_fwd(nhwc(dy, dtype), None, wd, C1 + C2, want_stats=False, split=C1)
        rd = F.conv_transpose2d(dy, w, None, 1, 1)
        tol = {BF: 1.2e-2, torch.float16: 2e-3}[dtype]
        assert_close(d1, rd[:, :C1], tol, f"{name} dgrad split 1")
        assert_close(d2, rd[:, C1:], tol, f"{name} dgrad split 2")


@pytest.mark.parametrize("layer", [l for l in C2_LAYERS if l[0] in ("Conv1b", "Conv3a", "Conv4b", "Conv5b", "Up_conv5a", "Up4", "Up2")],
                         ids=lambda l: l[0])
def test_c2_layer_f32(layer):
    _run_layer(16, layer, torch.float32, seed=77)


@pytest.mark.parametrize("layer", [l for l in C2_LAYERS if l[0] in cl.ENCODER], ids=lambda l: l[0])
def test_c2_encoder_wgrad_pair(layer):
    """cy_conv3x3_wgrad_pair at the step's own sizes: the labeled pass's 16 images and the unlabeled pass's 32
    (own tensors, own BN coefficients) in one launch = the gradient over all 48"""
    from cyhip import ops
    name, H, C1, C2, Cout, mode, pro = layer
    segs, ref = [], None
    for n, seed in ((16, 31), (32, 32)):
        x1, x2, w, dy, scale, shift = _case(n, layer, BF, seed)
        wv = w.clone().requires_grad_(True)
        (F.conv2d(_conv_input(x1, x2, mode, scale, shift, BF), wv, None, 1, 1) * dy).sum().backward()
        ref = wv.grad if ref is None else ref + wv.grad
        segs.append((nhwc(x1, BF), None, nhwc(dy, BF), None if scale is None else scale.to(DEV),
                     None if shift is None else shift.to(DEV)))
    out = torch.zeros(Cout, C1 + C2, 3, 3, device=DEV)
    (a1, a2, ady, asc, ash), (b1, b2, bdy, bsc, bsh) = segs
    ops.conv3x3_wgrad_pair(a1, a2, ady, asc, ash, b1, b2, bdy, bsc, bsh, mode=mode, out=out)
    assert_close(out, ref, 2e-3, f"pair wgrad {name}")


C4_LAYERS = cl.unet_layers(256, 512)


@pytest.mark.parametrize("layer", C4_LAYERS, ids=[l[0] for l in C4_LAYERS])
def test_c4_layer_bf16_igemm_path(layer):
    """BASELINE config 4 geometry: 256 x 256 inputs (widths 256..16: the igemm kernel), 4 images"""
    _run_layer(4, layer, BF, seed=4000, expect_kernel=("conv3x3_igemm_kernel", "conv3x3_flow_kernel"))


def test_full_size_c2_step_f32_matches_oracle():
    """one SemiSupervisedEpocher + InfoNCE step at the benchmark's own size (16 labeled + 16 unlabeled slices
    of 224 x 224, UNet(max_channel=512)) in f32 verification mode against oracle/step.py: losses and logits
    within 1e-4 (BASELINE.json north_star)"""
    from tests.step_harness import compare_step_with_oracle
    r = compare_step_with_oracle(n_l=16, n_unl=16, hw=224, max_channel=512, dtype=torch.float32, py_seed=3)
    assert r["rel_sup"] < 1e-4 and r["rel_reg"] < 1e-4 and r["rel_total"] < 1e-4, r
    assert r["rel_logits"] < 1e-4 and r["rel_logits_tf"] < 1e-4, r
    assert r["rel_running_mean"] < 1e-4, r
