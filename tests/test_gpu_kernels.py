"""Parity of every HIP kernel family (called through the C ABI via cyhip.ops) against the
oracle / stock torch CPU ops on the same seeded inputs.  f32 kernels: tight tolerances
(f32 MFMA = exact fmaf chains); bf16 kernels: inputs are bf16-representable, arithmetic is
f32-accumulate, so the only differences are accumulation order and the final bf16 rounding
(tolerance 1e-2 of the tensor's max magnitude, stated per test)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from cyhip import ops
    return ops


def nhwc(t, dtype=None):
    """CPU NCHW tensor -> GPU tensor with NHWC memory"""
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def cpu(t):
    return t.detach().float().cpu()


def rnd(*shape, gen, scale=1.0):
    return (torch.rand(*shape, generator=gen) * 2 - 1) * scale


def assert_close(a, b, rel, what=""):
    a, b = cpu(a).double(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} > {rel:.1e} * {scale:.3e}"


TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2, torch.float16: 2e-3}
WTOL = {torch.float32: 2e-5, torch.bfloat16: 2e-3, torch.float16: 5e-4}   # f32 weight gradients
ATOL = {torch.float32: 1e-5, torch.bfloat16: 1e-2, torch.float16: 2e-3}   # elementwise activations
DTYPES = [torch.float32, torch.bfloat16, torch.float16]

CONV_CASES = [
    # N, H, W, C1, C2, Cout, mode, prologue
    (2, 32, 32, 32, 0, 32, 0, 0),     # 8x32 tile, BN 32, small-K (bf16)
    (2, 32, 32, 64, 0, 64, 0, 1),     # 8x32 tile, BN 64, prologue
    (1, 16, 64, 64, 64, 128, 0, 0),   # 8x32 tile, BN 128, concat
    (2, 16, 16, 16, 0, 8, 0, 0),      # 16x16 tile, tiny channels (padding paths)
    (2, 16, 16, 32, 0, 64, 1, 0),     # 16x16 tile, pool on load
    (2, 16, 16, 128, 0, 256, 2, 0),   # 16x16 tile, upsample on load, BN 128 two cout tiles
    (3, 24, 8, 64, 0, 128, 0, 0),     # 32x8 tile, rows not a multiple of TH, spans images
    (2, 28, 28, 64, 64, 128, 0, 0),   # 8x28 tile (7 MFMA tiles), concat
    (3, 14, 14, 128, 0, 256, 0, 1),   # 16x14 tile spanning images, prologue
    (2, 14, 14, 32, 0, 32, 1, 0),     # generic masked 16x16 tile on 14x14, pool on load
    (1, 20, 20, 40, 24, 48, 0, 0),    # generic masked tile, odd channel counts, concat
    (2, 14, 14, 256, 0, 128, 0, 0),   # K = 4 chunks
    # widths that are multiples of 14 run the plane kernel (cy_conv_plane.h, 16 x 14 tiles):
    (2, 56, 56, 32, 0, 32, 0, 0),     # 32 couts, one chunk, weights of all taps resident
    (1, 28, 28, 64, 64, 64, 0, 0),    # 64 couts, 4 chunks with register-prefetched halo, concat
    (2, 28, 28, 16, 0, 8, 0, 0),      # tiny channels (padding paths)
    (1, 28, 56, 128, 0, 64, 2, 0),    # upsample on load, non-square
    (3, 14, 28, 64, 0, 48, 0, 1),     # tile spans images, rows not a multiple of 16, prologue
    (1, 28, 28, 64, 0, 96, 1, 0),     # pool on load (staged synchronously), 96 couts
    (5, 14, 14, 128, 128, 128, 0, 0), # 128 couts, concat, image borders inside every tile
]


def conv_ref(x1, x2, w, mode, scale, shift):
    a = x1
    if scale is not None:
        a = F.relu(a * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    if mode == 1:
        a = F.max_pool2d(a, 2, 2)
    elif mode == 2:
        a = F.interpolate(a, scale_factor=2, mode="nearest")
    if x2 is not None:
        a = torch.cat((a, x2), 1)
    return F.conv2d(a, w, None, 1, 1), a


def make_conv_case(case, dtype, seed):
    N, H, W, C1, C2, Cout, mode, pro = case
    g = torch.Generator().manual_seed(seed)
    sh, sw = (2 * H, 2 * W) if mode == 1 else ((H // 2, W // 2) if mode == 2 else (H, W))
    x1 = rnd(N, C1, sh, sw, gen=g).to(dtype).float()
    x2 = rnd(N, C2, H, W, gen=g).to(dtype).float() if C2 else None
    w = (rnd(Cout, C1 + C2, 3, 3, gen=g) / math.sqrt(9 * (C1 + C2))).to(dtype).float()
    scale = shift = None
    if pro:
        scale = rnd(C1, gen=g) + 0.2
        shift = rnd(C1, gen=g) * 0.3
    return x1, x2, w, scale, shift


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_fwd_and_stats(case, dtype):
    ops = _ops()
    N, H, W, C1, C2, Cout, mode, pro = case
    x1, x2, w, scale, shift = make_conv_case(case, dtype, 1)
    if pro and dtype != torch.float32:
        # the kernel rounds relu(scale*x+shift) to the storage type before the MFMA: do the same
        a = F.relu(x1 * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dtype).float()
        ref = F.conv2d(a, w, None, 1, 1)
    else:
        ref, _ = conv_ref(x1, x2, w, mode, scale, shift)
    wf, _ = ops.pack_weights(w.to(DEV), dtype)
    out, stats = ops.conv3x3_fwd(nhwc(x1, dtype), None if x2 is None else nhwc(x2, dtype), wf, Cout,
                                 mode=mode, scale=None if scale is None else scale.to(DEV),
                                 shift=None if shift is None else shift.to(DEV))
    torch.cuda.synchronize()
    assert_close(out, ref, TOL[dtype], f"conv {case} {dtype}")
    # statistics partials: sums of the (rounded) outputs the kernel itself stored
    o = cpu(out).double()
    s = cpu(stats).double().sum(0)
    count = N * H * W
    assert_close(s[0] / count, o.sum(dim=(0, 2, 3)) / count, 1e-4, "stat sum")
    assert_close(s[1] / count, (o * o).sum(dim=(0, 2, 3)) / count, 1e-4, "stat sumsq")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 32, 64, 64), (2, 14, 14, 64, 64, 256), (1, 28, 28, 128, 128, 128)])
def test_conv3x3_dgrad_with_split(shape, dtype):
    """data gradient with the two-destination (concat) epilogue, incl. the split-K path"""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    N, H, W, Cin1, Cin2, Cout = shape
    w = (rnd(Cout, Cin1 + Cin2, 3, 3, gen=g) / 20).to(dtype).float()
    dy = rnd(N, Cout, H, W, gen=g).to(dtype).float()
    ref = F.conv_transpose2d(dy, w, None, 1, 1)  # = dgrad of a stride-1 pad-1 conv
    _, wd = ops.pack_weights(w.to(DEV), dtype)
    (d1, d2), _ = ops.conv3x3_fwd(nhwc(dy, dtype), None, wd, Cin1 + Cin2, want_stats=False, split=Cin1)
    assert_close(d1, ref[:, :Cin1], TOL[dtype], "dgrad part 1")
    assert_close(d2, ref[:, Cin1:], TOL[dtype], "dgrad part 2")
    full, _ = ops.conv3x3_fwd(nhwc(dy, dtype), None, wd, Cin1 + Cin2, want_stats=False)
    assert_close(full, ref, TOL[dtype], "dgrad full")


WGRAD_CASES = [
    (2, 32, 32, 32, 0, 32, 0, 0),
    (2, 16, 16, 64, 0, 64, 0, 1),
    (1, 16, 32, 64, 64, 32, 0, 0),
    (2, 16, 16, 32, 0, 64, 1, 0),
    (2, 16, 16, 64, 0, 32, 2, 0),
    (3, 14, 14, 128, 0, 64, 0, 0),
    (2, 28, 28, 16, 0, 8, 0, 0),
    (1, 20, 20, 40, 24, 48, 0, 0),
    (2, 56, 56, 32, 0, 32, 0, 0),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv3x3_wgrad(case, dtype):
    ops = _ops()
    N, H, W, C1, C2, Cout, mode, pro = case
    x1, x2, w, scale, shift = make_conv_case(case, dtype, 3)
    g = torch.Generator().manual_seed(9)
    dy = rnd(N, Cout, H, W, gen=g).to(dtype).float()
    wv = w.clone().requires_grad_(True)
    if pro and dtype != torch.float32:
        a = F.relu(x1 * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dtype).float()
        y = F.conv2d(a, wv, None, 1, 1)
    else:
        y, _ = conv_ref(x1, x2, wv, mode, scale, shift)
    (y * dy).sum().backward()
    dw = ops.conv3x3_wgrad(nhwc(x1, dtype), None if x2 is None else nhwc(x2, dtype), nhwc(dy, dtype),
                           mode=mode, scale=None if scale is None else scale.to(DEV),
                           shift=None if shift is None else shift.to(DEV))
    assert_close(dw, wv.grad, WTOL[dtype], f"wgrad {case} {dtype}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout", [(1, 32), (3, 8), (1, 16)])
def test_first_conv_fwd_wgrad(cin, cout, dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    x = torch.rand(3, cin, 24, 40, generator=g)
    w = (rnd(cout, cin, 3, 3, gen=g) / 3).requires_grad_(True)
    ref = F.conv2d(x, w, None, 1, 1)
    out, stats = ops.conv_first_fwd(x.to(DEV), w.detach().to(DEV), dtype)
    assert_close(out, ref, TOL[dtype], "first conv")
    o = cpu(out).double()
    s = cpu(stats).double().sum(0)
    assert_close(s[0], o.sum(dim=(0, 2, 3)), 1e-4, "first conv stat sum")
    assert_close(s[1], (o * o).sum(dim=(0, 2, 3)), 1e-4, "first conv stat sumsq")
    dy = rnd(3, cout, 24, 40, gen=g).to(dtype).float()
    (ref * dy).sum().backward()
    dw = ops.conv_first_wgrad(x.to(DEV), nhwc(dy, dtype))
    assert_close(dw, w.grad, WTOL[dtype], "first conv wgrad")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,W", [(3, 24, 48), (2, 224, 224), (5, 8, 16), (1, 40, 2048)])
def test_first_conv_wgrad_on_the_matrix_core(N, H, W, dtype):
    """first_wgrad_mfma_kernel (Cin 1, Cout 32, 16-bit storage, H % 8 == 0, W % 16 == 0): rows = couts, columns = taps,
    k = 16 consecutive pixels; the image is rounded to the storage type like the forward kernel's.  Against torch's CPU
    convolution on the rounded image (exact products, f32 sums: 1e-4 of the maximum), and bit-identical over launches;
    image-border rows / columns, sub-block borders and several sub-blocks per workgroup are in these geometries."""
    ops = _ops()
    g = torch.Generator().manual_seed(N * H + W)
    x = torch.rand(N, 1, H, W, generator=g)
    w = (rnd(32, 1, 3, 3, gen=g) / 3).requires_grad_(True)
    dy = rnd(N, 32, H, W, gen=g).to(dtype).float()
    (F.conv2d(x.to(dtype).float(), w, None, 1, 1) * dy).sum().backward()
    dw = ops.conv_first_wgrad(x.to(DEV), nhwc(dy, dtype))
    assert_close(dw, w.grad, 1e-4, "first conv wgrad (matrix core)")
    dw2 = ops.conv_first_wgrad(x.to(DEV), nhwc(dy, dtype))
    assert torch.equal(dw, dw2)
    sink = torch.ones(32, 1, 3, 3, device=DEV)
    ops.conv_first_wgrad(x.to(DEV), nhwc(dy, dtype), out=sink)
    assert_close(sink - 1, w.grad, 1e-4, "accumulating form")


def test_first_conv_at_pretraining_batch_size():
    """BASELINE config 5 runs the first layer on 512 slices per launch: more image rows per workgroup than the
    LDS image of the input holds unless the grid grows with the batch (it did not at first: CY_ERR_SHAPE in
    `bench.py --workload c5`).  Forward + statistics and weight gradient on 480 slices against the same kernels
    on chunks of 60"""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    N, H, W, Cout = 480, 224, 224, 8
    x = torch.rand(N, 1, H, W, generator=g).to(DEV)
    w = (rnd(Cout, 1, 3, 3, gen=g) * 0.3).to(DEV)
    BF = torch.bfloat16
    out, stats = ops.conv_first_fwd(x, w, BF)
    dy = out  # any tensor of the right shape
    dw = ops.conv_first_wgrad(x, dy)
    s = stats.double().sum(0)
    s_ref = torch.zeros_like(s)
    dw_ref = torch.zeros_like(dw, dtype=torch.float64)
    for i in range(0, N, 60):
        o, st = ops.conv_first_fwd(x[i:i + 60].contiguous(), w, BF)
        assert torch.equal(o, out[i:i + 60])
        s_ref += st.double().sum(0)
        dw_ref += ops.conv_first_wgrad(x[i:i + 60].contiguous(), o.contiguous(memory_format=torch.channels_last)).double()
    assert_close(s, s_ref.cpu(), 1e-6, "statistics")
    assert_close(dw, dw_ref.cpu(), 1e-5, "first-layer wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,hw", [(32, 24), (8, 12), (512, 6), (40, 10)])
def test_bn_relu_fwd_bwd(C, hw, dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    N = 3
    y = (rnd(N, C, hw, hw, gen=g) * 2 + 0.3).to(dtype).float().requires_grad_(True)
    gamma = (rnd(C, gen=g) + 1.2).requires_grad_(True)
    beta = (rnd(C, gen=g) * 0.5).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    out_ref = F.relu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.01, 1e-5))
    da = rnd(N, C, hw, hw, gen=g).to(dtype).float()
    (out_ref * da).sum().backward()
    # statistics exactly as the conv epilogue would emit them (one partial here)
    yd = y.detach().double()
    part = torch.stack([yd.sum(dim=(0, 2, 3)), (yd * yd).sum(dim=(0, 2, 3))]).float().view(1, 2, C).to(DEV)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    count = N * hw * hw
    scale, shift, mean, invstd = ops.bn_finalize(part, count, gamma.detach().to(DEV), beta.detach().to(DEV),
                                                 rmd, rvd, 0.01, 1e-5, True, True, C, DEV)
    assert_close(rmd, rm, 1e-5, "running mean")
    assert_close(rvd, rv, 1e-5, "running var")
    yg = nhwc(y.detach(), dtype)
    out = ops.bn_relu_apply(yg, scale, shift)
    assert_close(out, out_ref, ATOL[dtype], "bn relu apply")
    # the pooled form: the same block output bit for bit, and MaxPool2d(2) of exactly those stored values
    out_p, pooled = ops.bn_relu_apply_pool(yg, scale, shift)
    assert torch.equal(out_p, out)
    assert torch.equal(pooled.float().cpu(), F.max_pool2d(out.float().cpu(), 2, 2))
    assert pooled.is_contiguous(memory_format=torch.channels_last) and pooled.dtype == out.dtype
    dy, dgamma, dbeta = ops.bn_relu_bwd(nhwc(da, dtype), yg, scale, shift, mean, invstd, True)
    assert_close(dgamma, gamma.grad, 1e-4, "dgamma")
    assert_close(dbeta, beta.grad, 1e-4, "dbeta")
    assert_close(dy, y.grad, 1e-4 if dtype == torch.float32 else TOL[dtype], "bn bwd dy")
    # eval-mode scale/shift from running stats, no update
    rm2, rv2 = rmd.clone(), rvd.clone()
    s2, h2, _, _ = ops.bn_finalize(None, count, gamma.detach().to(DEV), beta.detach().to(DEV), rm2, rv2, 0.01,
                                   1e-5, False, False, C, DEV)
    ref_eval = F.relu(F.batch_norm(y.detach(), cpu(rmd), cpu(rvd), gamma.detach(), beta.detach(), False, 0.0, 1e-5))
    assert_close(ops.bn_relu_apply(yg, s2, h2), ref_eval, ATOL[dtype], "bn eval")
    assert torch.equal(rm2, rmd) and torch.equal(rv2, rvd)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_upsample_bwd(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = F.relu(rnd(2, 16, 12, 20, gen=g)).to(dtype).float().requires_grad_(True)  # zeros -> ties
    p = F.max_pool2d(x, 2, 2)
    dp = rnd(2, 16, 6, 10, gen=g).to(dtype).float()
    (p * dp).sum().backward()
    dx = ops.maxpool2_bwd(nhwc(x.detach(), dtype), nhwc(dp, dtype))
    assert_close(dx, x.grad, 1e-6, "maxpool bwd")
    add = rnd(2, 16, 12, 20, gen=g).to(dtype).float()
    dx2 = ops.maxpool2_bwd(nhwc(x.detach(), dtype), nhwc(dp, dtype), nhwc(add, dtype))
    assert_close(dx2, x.grad + add, 1e-6 if dtype == torch.float32 else ATOL[dtype], "maxpool bwd + add")
    u = rnd(2, 16, 6, 10, gen=g).requires_grad_(True)
    du = rnd(2, 16, 12, 20, gen=g).to(dtype).float()
    (F.interpolate(u, scale_factor=2, mode="nearest") * du).sum().backward()
    assert_close(ops.upsample2_bwd(nhwc(du, dtype)), u.grad, 1e-6 if dtype == torch.float32 else ATOL[dtype], "upsample bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,hw,N", [(32, 24, 3), (64, 12, 2), (512, 4, 5), (40, 8, 2)])
def test_pool_bwd_with_bn_backward_sums(C, hw, N, dtype):
    """maxpool2_bwd_bn = maxpool2_bwd (+ addend) whose second output, fed to the BatchNorm backward instead of its own
    reduction pass, gives the same dy / dgamma / dbeta (C = 40: five channel groups do not divide a workgroup, the
    unfused form runs)"""
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    y = nhwc(rnd(N, C, hw, hw, gen=g) * 1.5, dtype)
    gamma, beta = rnd(C, gen=g) + 1.2, rnd(C, gen=g) * 0.5
    yd = cpu(y).double()
    part = torch.stack([yd.sum(dim=(0, 2, 3)), (yd * yd).sum(dim=(0, 2, 3))]).float().view(1, 2, C).to(DEV)
    scale, shift, mean, invstd = ops.bn_finalize(part, N * hw * hw, gamma.to(DEV), beta.to(DEV), torch.zeros(C, device=DEV),
                                                 torch.ones(C, device=DEV), 0.1, 1e-5, True, True, C, DEV)
    out, _ = ops.bn_relu_apply_pool(y, scale, shift)
    dpool = nhwc(rnd(N, C, hw // 2, hw // 2, gen=g), dtype)
    for add in (None, nhwc(rnd(N, C, hw, hw, gen=g), dtype)):
        dx_ref = ops.maxpool2_bwd(out, dpool, add)
        dx, partials = ops.maxpool2_bwd_bn(out, dpool, add, y, scale, shift, mean, invstd)
        assert torch.equal(dx, dx_ref)
        assert (partials is None) == (C == 40)
        ref = ops.bn_relu_bwd(dx_ref, y, scale, shift, mean, invstd, True)
        got = ops.bn_relu_bwd(dx, y, scale, shift, mean, invstd, True, partials=partials)
        for a, b, what in zip(got, ref, ("dy", "dgamma", "dbeta")):
            assert_close(a, cpu(b), 2e-5 if what != "dy" else (1e-5 if dtype == torch.float32 else TOL[dtype]), what)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("K", [4, 2, 5, 8])
def test_head_and_losses(K, dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    N, C, H, W = 2, 32, 16, 24
    x = rnd(N, C, H, W, gen=g).to(dtype).float().requires_grad_(True)
    w = (rnd(K, C, 1, 1, gen=g) / 4).requires_grad_(True)
    b = (rnd(K, gen=g) / 4).requires_grad_(True)
    tgt = torch.randint(0, K, (N, H, W), generator=g)
    logits_ref = F.conv2d(x, w, b)
    from oracle.losses import sup_loss, softmax_mse
    loss_ref = sup_loss(logits_ref, tgt)
    loss_ref.backward()
    xg = nhwc(x.detach(), dtype)
    logits = ops.head_fwd(xg, w.detach().to(DEV), b.detach().to(DEV))
    assert_close(logits, logits_ref, 1e-5, "head fwd")
    loss = ops.softmax_kl_fwd(logits, tgt.to(DEV), 1e-16)
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * max(1.0, abs(loss_ref.item()))
    gs = torch.ones(1, device=DEV)
    dl = ops.softmax_kl_bwd(logits, tgt.to(DEV), gs, 1e-16)
    dx, dw, db = ops.head_bwd(xg, w.detach().to(DEV), dl, True, True)
    assert_close(dw, w.grad, 1e-4, "head dw")
    assert_close(db, b.grad, 1e-4, "head db")
    assert_close(dx, x.grad, 1e-4 if dtype == torch.float32 else ATOL[dtype], "head dx")
    # softmax-MSE pair
    a = rnd(N, K, H, W, gen=g).requires_grad_(True)
    c = rnd(N, K, H, W, gen=g).requires_grad_(True)
    m_ref = softmax_mse(a, c)
    m_ref.backward()
    ag, cg = nhwc(a.detach()), nhwc(c.detach())
    m = ops.softmax_mse_fwd(ag, cg)
    assert abs(m.item() - m_ref.item()) < 1e-6
    da, dc = ops.softmax_mse_bwd(ag, cg, gs, True, True)
    assert_close(da, a.grad, 1e-4, "mse da")
    assert_close(dc, c.grad, 1e-4, "mse db")
    # dice counts
    counts = cpu(ops.dice_counts(logits, tgt.to(DEV))).long()
    pred = logits_ref.detach().argmax(1)
    for n in range(N):
        for k in range(K):
            inter = ((pred[n] == k) & (tgt[n] == k)).sum().item()
            union = (pred[n] == k).sum().item() + (tgt[n] == k).sum().item()
            assert counts[n, k, 0].item() == inter and counts[n, k, 1].item() == union


def test_projector_pieces(golden_dir):
    ops = _ops()
    from oracle import losses as ol
    gld = np.load(golden_dir / "heads_losses.npz")
    psd = ol.init_projector_sd(128, 256, 256, seed=3)
    feat = torch.from_numpy(gld["proj_feat"])
    fg = nhwc(feat)
    pooled = ops.avgpool_fwd(fg)
    assert_close(pooled, feat.mean(dim=(2, 3)), 1e-6, "avgpool")
    h = ops.linear_fwd(pooled, psd["_header.2.weight"].to(DEV), psd["_header.2.bias"].to(DEV), 1, 0.01)
    o = ops.linear_fwd(h, psd["_header.4.weight"].to(DEV), psd["_header.4.bias"].to(DEV), 0, 0.0)
    z, norms = ops.l2norm_fwd(o)
    assert_close(z, torch.from_numpy(gld["proj_z"]), 2e-5, "projection head output (golden)")
    # backward chain against the golden gradients of the reference module
    gz = torch.linspace(-1, 1, z.numel()).view_as(z).to(DEV)
    do = ops.l2norm_bwd(o, norms, gz)
    dh, dw4, db4 = ops.linear_bwd(h, psd["_header.4.weight"].to(DEV), o, do, 0, 0.0, True, True)
    dp, dw2, db2 = ops.linear_bwd(pooled, psd["_header.2.weight"].to(DEV), h, dh, 1, 0.01, True, True)
    dfeat = ops.avgpool_bwd(dp, tuple(feat.shape), torch.float32)
    assert_close(dw4, torch.from_numpy(gld["proj_grad__header.4.weight"]), 1e-4, "dW4")
    assert_close(db4, torch.from_numpy(gld["proj_grad__header.4.bias"]), 1e-4, "db4")
    assert_close(dw2, torch.from_numpy(gld["proj_grad__header.2.weight"]), 1e-4, "dW2")
    assert_close(db2, torch.from_numpy(gld["proj_grad__header.2.bias"]), 1e-4, "db2")
    assert_close(dfeat, torch.from_numpy(gld["proj_dfeat"]), 1e-4, "dfeat")


def test_projection_head_one_launch(golden_dir):
    """cy_proj_head_fwd / bwd (the whole ProjectionHead per direction) against the reference's own vectors, and at
    the encoder hook's size (96 x 512 x 14 x 14, bf16) against the separate kernels"""
    ops = _ops()
    from oracle import losses as ol
    gld = np.load(golden_dir / "heads_losses.npz")
    psd = {k: v.to(DEV) for k, v in ol.init_projector_sd(128, 256, 256, seed=3).items()}
    w1, b1, w2, b2 = (psd["_header.2.weight"], psd["_header.2.bias"], psd["_header.4.weight"], psd["_header.4.bias"])
    feat = torch.from_numpy(gld["proj_feat"])
    fg = nhwc(feat)
    z, pooled, y1, y2, norms = ops.proj_head_fwd(fg, w1, b1, w2, b2)
    assert_close(z, torch.from_numpy(gld["proj_z"]), 2e-5, "z (golden)")
    gz = torch.linspace(-1, 1, z.numel()).view_as(z).to(DEV)
    dfeat, (dw1, db1, dw2, db2) = ops.proj_head_bwd(gz, pooled, y1, y2, norms, w1, w2, tuple(feat.shape),
                                                    torch.float32, True)
    assert_close(dw2, torch.from_numpy(gld["proj_grad__header.4.weight"]), 1e-4, "dW (second linear)")
    assert_close(db2, torch.from_numpy(gld["proj_grad__header.4.bias"]), 1e-4, "db (second linear)")
    assert_close(dw1, torch.from_numpy(gld["proj_grad__header.2.weight"]), 1e-4, "dW (first linear)")
    assert_close(db1, torch.from_numpy(gld["proj_grad__header.2.bias"]), 1e-4, "db (first linear)")
    assert_close(dfeat, torch.from_numpy(gld["proj_dfeat"]), 1e-4, "dfeat")
    # accumulation into existing buffers
    sinks = tuple(torch.ones_like(t) for t in (dw1, db1, dw2, db2))
    ops.proj_head_bwd(gz, pooled, y1, y2, norms, w1, w2, tuple(feat.shape), torch.float32, False, sinks)
    for got, ref in zip(sinks, (dw1, db1, dw2, db2)):
        assert_close(got - 1, ref, 1e-5, "added into")

    g = torch.Generator().manual_seed(21)
    B, C = 96, 512
    psd = {k: v.to(DEV) for k, v in ol.init_projector_sd(C, 256, 256, seed=4).items()}
    w1, b1, w2, b2 = (psd["_header.2.weight"], psd["_header.2.bias"], psd["_header.4.weight"], psd["_header.4.bias"])
    x = nhwc(rnd(B, C, 14, 14, gen=g), torch.bfloat16)
    z, pooled, y1, y2, norms = ops.proj_head_fwd(x, w1, b1, w2, b2)
    p0 = ops.avgpool_fwd(x)
    h0 = ops.linear_fwd(p0, w1, b1, 1, 0.01)
    o0 = ops.linear_fwd(h0, w2, b2, 0, 0.0)
    z0, n0 = ops.l2norm_fwd(o0)
    assert_close(z, cpu(z0), 1e-5, "z vs separate kernels")
    gz = rnd(B, 256, gen=g).to(DEV)
    dx, (dw1, db1, dw2, db2) = ops.proj_head_bwd(gz, pooled, y1, y2, norms, w1, w2, tuple(x.shape), torch.bfloat16,
                                                 True)
    do = ops.l2norm_bwd(o0, n0, gz)
    dh, rw2, rb2 = ops.linear_bwd(h0, w2, o0, do, 0, 0.0, True, True)
    dp, rw1, rb1 = ops.linear_bwd(p0, w1, h0, dh, 1, 0.01, True, True)
    rx = ops.avgpool_bwd(dp, tuple(x.shape), torch.bfloat16)
    assert_close(dw2, cpu(rw2), 1e-5, "dW2"), assert_close(db2, cpu(rb2), 1e-5, "db2")
    assert_close(dw1, cpu(rw1), 1e-5, "dW1"), assert_close(db1, cpu(rb1), 1e-5, "db1")
    assert_close(dx, cpu(rx), 1e-2, "dx (bf16)")


def test_sgemm():
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    A, B = rnd(70, 45, gen=g), rnd(100, 45, gen=g)
    assert_close(ops.sgemm(A.to(DEV), B.to(DEV), 0.5, True), 0.5 * A @ B.t(), 1e-6, "sgemm NT")
    B2 = rnd(45, 33, gen=g)
    assert_close(ops.sgemm(A.to(DEV), B2.to(DEV), 2.0, False), 2.0 * A @ B2, 1e-6, "sgemm NN")


def test_supcon_against_reference_goldens(golden_dir):
    ops = _ops()
    gld = np.load(golden_dir / "heads_losses.npz")
    z1, z2 = torch.from_numpy(gld["sc_z1"]), torch.from_numpy(gld["sc_z2"])
    P = torch.cat([z1, z2]).to(DEV)
    n = z1.shape[0]
    gs = torch.ones(1, device=DEV)
    for tag, target in (("simclr", list(range(n))), ("partition", [0, 1, 2, 0, 1, 2, 0, 1]),
                        ("patient", [0, 1, 2, 3, 3, 4, 5, 6])):
        lab = torch.tensor(target, dtype=torch.int32, device=DEV)
        loss, S, stats = ops.supcon_fwd(P, lab, None, 0.07)
        assert abs(loss.item() - float(gld[f"sc_{tag}_loss"])) < 1e-5 * abs(float(gld[f"sc_{tag}_loss"])), tag
        dP = ops.supcon_bwd(P, lab, None, S, stats, gs, 0.07)
        assert_close(dP[:n], torch.from_numpy(gld[f"sc_{tag}_dz1"]), 2e-4, f"{tag} dz1")
        assert_close(dP[n:], torch.from_numpy(gld[f"sc_{tag}_dz2"]), 2e-4, f"{tag} dz2")
        sl, se, po, ne = ops.supcon_matrices(S, stats, lab, None)
        assert_close(sl, torch.from_numpy(gld[f"sc_{tag}_sim_logits"]), 1e-5, "sim_logits")
        assert_close(se, torch.from_numpy(gld[f"sc_{tag}_sim_exp"]), 1e-5, "sim_exp")
        assert_close(po, torch.from_numpy(gld[f"sc_{tag}_pos"]), 0, "pos_mask")
        assert_close(ne, torch.from_numpy(gld[f"sc_{tag}_neg"]), 0, "neg_mask")
    pm = torch.from_numpy(gld["sc_mask"]).to(torch.uint8).to(DEV)
    loss, S, stats = ops.supcon_fwd(P, None, pm, 0.07)
    assert abs(loss.item() - float(gld["sc_mask_loss"])) < 1e-5 * abs(float(gld["sc_mask_loss"]))
    dP = ops.supcon_bwd(P, None, pm, S, stats, gs, 0.07)
    assert_close(dP[:n], torch.from_numpy(gld["sc_mask_dz1"]), 2e-4, "mask dz1")


def test_supcon_large_against_oracle():
    """512 rows (C5-style) vs the oracle in f64."""
    ops = _ops()
    from oracle.losses import supcon_loss
    g = torch.Generator().manual_seed(10)
    n, D = 256, 256
    z1 = F.normalize(torch.randn(n, D, generator=g), dim=1)
    z2 = F.normalize(z1 + 0.3 * torch.randn(n, D, generator=g), dim=1)
    target = [i % 3 for i in range(n)]
    a, b = z1.double().requires_grad_(True), z2.double().requires_grad_(True)
    ref = supcon_loss(a, b, target=target)
    ref.backward()
    P = torch.cat([z1, z2]).to(DEV)
    lab = torch.tensor(target, dtype=torch.int32, device=DEV)
    loss, S, stats = ops.supcon_fwd(P, lab, None, 0.07)
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item())
    dP = ops.supcon_bwd(P, lab, None, S, stats, torch.ones(1, device=DEV), 0.07)
    assert_close(dP, torch.cat([a.grad, b.grad]).float(), 1e-4, "large supcon dP")


def test_supcon_c5_size_against_oracle():
    """BASELINE config 5: 4096 global embeddings (2 x 2048 rows, D = 256), partition labels, f64 oracle"""
    ops = _ops()
    from oracle.losses import supcon_loss
    g = torch.Generator().manual_seed(11)
    n, D = 2048, 256
    z1 = F.normalize(torch.randn(n, D, generator=g), dim=1)
    z2 = F.normalize(z1 + 0.5 * torch.randn(n, D, generator=g), dim=1)
    target = [i % 3 for i in range(n)]
    a, b = z1.double().requires_grad_(True), z2.double().requires_grad_(True)
    ref = supcon_loss(a, b, target=target)
    ref.backward()
    P = torch.cat([z1, z2]).to(DEV)
    lab = torch.tensor(target, dtype=torch.int32, device=DEV)
    loss, S, stats = ops.supcon_fwd(P, lab, None, 0.07)
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()), (loss.item(), ref.item())
    dP = ops.supcon_bwd(P, lab, None, S, stats, torch.ones(1, device=DEV), 0.07)
    assert_close(dP, torch.cat([a.grad, b.grad]).float(), 1e-4, "C5 supcon dP")


def test_supcon_fused_against_reference_goldens(golden_dir):
    """the fused kernels (similarity / gradient matrices never materialised) against the reference's own vectors"""
    ops = _ops()
    gld = np.load(golden_dir / "heads_losses.npz")
    z1, z2 = torch.from_numpy(gld["sc_z1"]), torch.from_numpy(gld["sc_z2"])
    P = torch.cat([z1, z2]).to(DEV)
    assert ops.supcon_fused_ok(P)
    n = z1.shape[0]
    gs = torch.ones(1, device=DEV)
    for tag, target in (("simclr", list(range(n))), ("partition", [0, 1, 2, 0, 1, 2, 0, 1]),
                        ("patient", [0, 1, 2, 3, 3, 4, 5, 6])):
        lab = torch.tensor(target, dtype=torch.int32, device=DEV)
        loss, diag, stats = ops.supcon_fwd_fused(P, lab, None, 0.07)
        assert abs(loss.item() - float(gld[f"sc_{tag}_loss"])) < 1e-5 * abs(float(gld[f"sc_{tag}_loss"])), tag
        assert_close(diag, (P * P).sum(1).cpu() / 0.07, 1e-6, "diag")
        dP = ops.supcon_bwd_fused(P, lab, None, stats, gs, 0.07)
        assert_close(dP[:n], torch.from_numpy(gld[f"sc_{tag}_dz1"]), 2e-4, f"{tag} dz1")
        assert_close(dP[n:], torch.from_numpy(gld[f"sc_{tag}_dz2"]), 2e-4, f"{tag} dz2")
    pm = torch.from_numpy(gld["sc_mask"]).to(torch.uint8).to(DEV)
    loss, diag, stats = ops.supcon_fwd_fused(P, None, pm, 0.07)
    assert abs(loss.item() - float(gld["sc_mask_loss"])) < 1e-5 * abs(float(gld["sc_mask_loss"]))
    dP = ops.supcon_bwd_fused(P, None, pm, stats, gs, 0.07)
    assert_close(dP[:n], torch.from_numpy(gld["sc_mask_dz1"]), 2e-4, "mask dz1")


@pytest.mark.parametrize("n,D", [(256, 256), (2048, 256), (37, 64), (100, 128), (65, 24)])
def test_supcon_fused_against_oracle_and_unfused(n, D):
    """row counts that are not multiples of the 64-row tile, D below 256, the C5 size (4096 embeddings): loss and
    gradient against the f64 oracle, and against the kernels that materialise S; twice for reproducibility"""
    ops = _ops()
    from oracle.losses import supcon_loss
    g = torch.Generator().manual_seed(1000 + n)
    z1 = F.normalize(torch.randn(n, D, generator=g), dim=1)
    z2 = F.normalize(z1 + 0.4 * torch.randn(n, D, generator=g), dim=1)
    target = [i % 3 for i in range(n)]
    a, b = z1.double().requires_grad_(True), z2.double().requires_grad_(True)
    ref = supcon_loss(a, b, target=target)
    ref.backward()
    P = torch.cat([z1, z2]).to(DEV)
    lab = torch.tensor(target, dtype=torch.int32, device=DEV)
    gsc = torch.full((1,), 0.7, device=DEV)
    loss, diag, stats = ops.supcon_fwd_fused(P, lab, None, 0.07)
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()), (loss.item(), ref.item())
    dP = ops.supcon_bwd_fused(P, lab, None, stats, gsc, 0.07)
    assert_close(dP, 0.7 * torch.cat([a.grad, b.grad]).float(), 1e-4, "fused dP vs oracle")
    loss_u, S, stats_u = ops.supcon_fwd(P, lab, None, 0.07)
    assert abs(loss.item() - loss_u.item()) < 1e-6 * abs(loss_u.item())
    assert_close(stats[:, :3], cpu(stats_u[:, :3]), 1e-5, "row statistics")
    assert_close(dP, cpu(ops.supcon_bwd(P, lab, None, S, stats_u, gsc, 0.07)), 1e-5, "fused dP vs unfused")
    loss2, _, stats2 = ops.supcon_fwd_fused(P, lab, None, 0.07)
    dP2 = ops.supcon_bwd_fused(P, lab, None, stats2, gsc, 0.07)
    assert torch.equal(loss, loss2) and torch.equal(dP, dP2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_affine_fwd_bwd(dtype):
    ops = _ops()
    from oracle.losses import affine_nearest, make_theta
    g = torch.Generator().manual_seed(12)
    N, C, H, W = 4, 12, 14, 14
    thetas = torch.stack([make_theta(1.0, 0.0, 0.0, 0.0, False, False),
                          make_theta(1.25, 30.0, 0.08, -0.05, False, True),
                          make_theta(0.82, -41.0, -0.1, 0.1, True, False),
                          make_theta(1.1, 12.0, 0.02, 0.03, True, True)])
    x = rnd(N, C, H, W, gen=g).to(dtype).float().requires_grad_(True)
    ref = affine_nearest(x, thetas)
    dout = rnd(N, C, H, W, gen=g).to(dtype).float()
    (ref * dout).sum().backward()
    out = ops.affine_fwd(nhwc(x.detach(), dtype), thetas.to(DEV))
    assert_close(out, ref, 1e-6, "affine fwd")
    dx = ops.affine_bwd(nhwc(dout, dtype), thetas.to(DEV))
    assert_close(dx, x.grad, 1e-6 if dtype == torch.float32 else ATOL[dtype], "affine bwd")
    if dtype == torch.float32:
        img = torch.rand(N, 1, 32, 32, generator=g)
        gam = torch.tensor([0.5, 1.0, 1.7, 2.0])
        refi = affine_nearest(img, thetas, gam)
        outi = ops.affine_fwd(img.to(DEV), thetas.to(DEV), gam.to(DEV))
        assert_close(outi, refi, 1e-5, "affine + gamma image")


def test_ema_and_radam():
    ops = _ops()
    from oracle.losses import ema_update
    g = torch.Generator().manual_seed(13)
    t, s = rnd(5000, gen=g), rnd(5000, gen=g)
    tg = t.to(DEV)
    ops.ema_update(tg, s.to(DEV), 0.99, 1e-5)
    assert_close(tg, ema_update(t, s, 0.99, 1e-5), 1e-6, "ema")
    p = rnd(3000, gen=g).requires_grad_(True)
    opt = torch.optim.RAdam([p], lr=1e-3, weight_decay=1e-5)
    pg = p.detach().clone().to(DEV)
    m, v = torch.zeros_like(pg), torch.zeros_like(pg)
    for step in range(1, 9):
        grad = rnd(3000, gen=g)
        p.grad = grad.clone()
        opt.step()
        ops.radam_step(pg, grad.to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-5, step)
        assert_close(pg, p.detach(), 1e-6, f"radam step {step}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_batched_weight_pack_equals_per_layer_pack(dtype):
    """cy_conv3x3_pack_weights_batched (one launch for all layers) writes the same packed images as
    cy_conv3x3_pack_weights layer by layer"""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    shapes = [(32, 32), (64, 32), (8, 16), (128, 64), (48, 40), (256, 128)]
    ws = [rnd(co, ci, 3, 3, gen=g).to(DEV) for co, ci in shapes]
    packs = ops.pack_weights_batched(ws, dtype)
    again = ops.pack_weights_batched(ws, dtype)  # second call: cached layer table, fresh arenas
    for w, (wf, wd), (wf2, wd2) in zip(ws, packs, again):
        rf, rd = ops.pack_weights(w, dtype)
        assert torch.equal(wf, rf) and torch.equal(wd, rd), tuple(w.shape)
        assert torch.equal(wf2, rf) and torch.equal(wd2, rd), tuple(w.shape)


def _random_conv_cases(n, seed):
    """seeded random geometries for the conv kernels: widths that select the plane kernel (multiples
    of 14) and widths that do not, row counts that leave partial tiles, channel counts that are not
    multiples of the cout / chunk tiles, every load mode, concat and prologue"""
    import random as _r
    rng = _r.Random(seed)
    cases = []
    while len(cases) < n:
        W = rng.choice([14, 28, 42, 56, 16, 24, 32])
        H = rng.choice([14, 28, 16, 10, 6])
        N = rng.choice([1, 2, 3, 5])
        mode = rng.choice([0, 0, 0, 1, 2])
        if mode == 2 and (H % 2 or W % 2):
            continue
        C1 = 8 * rng.randint(1, 20)
        C2 = 0 if mode or rng.random() < 0.6 else 8 * rng.randint(1, 10)
        Cout = 8 * rng.randint(1, 24)
        pro = int(C2 == 0 and mode == 0 and rng.random() < 0.4)
        cases.append((N, H, W, C1, C2, Cout, mode, pro))
    return cases


@pytest.mark.parametrize("case", _random_conv_cases(24, seed=11))
def test_conv3x3_random_geometries_bf16(case):
    """forward + statistics, data gradient and weight gradient of one random layer against torch (bf16)"""
    ops = _ops()
    dtype = torch.bfloat16
    N, H, W, C1, C2, Cout, mode, pro = case
    x1, x2, w, scale, shift = make_conv_case(case, dtype, 17)
    g = torch.Generator().manual_seed(23)
    dy = rnd(N, Cout, H, W, gen=g).to(dtype).float()
    wv = w.clone().requires_grad_(True)
    if pro:
        a_in = F.relu(x1 * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dtype).float()
        ref = F.conv2d(a_in, wv, None, 1, 1)
    else:
        ref, a_in = conv_ref(x1, x2, wv, mode, scale, shift)
    (ref * dy).sum().backward()
    kw = dict(mode=mode, scale=None if scale is None else scale.to(DEV), shift=None if shift is None else shift.to(DEV))
    wf, wd = ops.pack_weights(w.to(DEV), dtype)
    g1, g2 = nhwc(x1, dtype), None if x2 is None else nhwc(x2, dtype)
    out, stats = ops.conv3x3_fwd(g1, g2, wf, Cout, **kw)
    assert_close(out, ref.detach(), TOL[dtype], f"fwd {case}")
    o = cpu(out).double()
    s = cpu(stats).double().sum(0)
    assert_close(s[0] / o.numel(), o.sum(dim=(0, 2, 3)) / o.numel(), 1e-4, f"stat sum {case}")
    assert_close(s[1] / o.numel(), (o * o).sum(dim=(0, 2, 3)) / o.numel(), 1e-4, f"stat sumsq {case}")
    # data gradient w.r.t. the conv's (concatenated, pooled / upsampled) input
    din, _ = ops.conv3x3_fwd(nhwc(dy, dtype), None, wd, C1 + C2, want_stats=False)
    assert_close(din, F.conv_transpose2d(dy, w, None, 1, 1), TOL[dtype], f"dgrad {case}")
    dw = ops.conv3x3_wgrad(g1, g2, nhwc(dy, dtype), **kw)
    assert_close(dw, wv.grad, 2e-3, f"wgrad {case}")


@pytest.mark.parametrize("case", [(2, 28, 28, 64, 0, 64, 0, 1), (1, 56, 56, 32, 0, 64, 1, 0),
                                  (3, 14, 14, 128, 0, 256, 0, 0), (2, 28, 28, 32, 32, 32, 0, 0)])
def test_conv3x3_wgrad_pair_equals_the_sum_of_two_launches(case):
    """cy_conv3x3_wgrad_pair: the same layer on two batches (own tensors, own BN coefficients) in one
    launch accumulates dw_a + dw_b"""
    ops = _ops()
    dtype = torch.bfloat16
    N, H, W, C1, C2, Cout, mode, pro = case
    segs = []
    for n, seed in ((N, 31), (N + 1, 32)):
        c = (n,) + case[1:]
        x1, x2, w, scale, shift = make_conv_case(c, dtype, seed)
        g = torch.Generator().manual_seed(seed + 100)
        dy = rnd(n, Cout, H, W, gen=g).to(dtype).float()
        segs.append((nhwc(x1, dtype), None if x2 is None else nhwc(x2, dtype), nhwc(dy, dtype),
                     None if scale is None else scale.to(DEV), None if shift is None else shift.to(DEV)))
    base = torch.randn(Cout, C1 + C2, 3, 3, device=DEV)
    ref = base.clone()
    for s1, s2, dy, sc, sh in segs:
        ops.conv3x3_wgrad(s1, s2, dy, mode=mode, scale=sc, shift=sh, out=ref)
    out = base.clone()
    (a1, a2, ady, asc, ash), (b1, b2, bdy, bsc, bsh) = segs
    ops.conv3x3_wgrad_pair(a1, a2, ady, asc, ash, b1, b2, bdy, bsc, bsh, mode=mode, out=out)
    assert_close(out - base, cpu(ref - base), 1e-5, f"pair wgrad {case}")
