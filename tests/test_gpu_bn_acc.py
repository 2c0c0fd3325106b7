"""BatchNorm sums on accumulators (csrc/cy_bn_acc.h, ABI v11): the producers add their partial sums with integer atomics,
the consumers derive the coefficients themselves, no finalize launch.  Reference semantics: nn.BatchNorm2d in training
mode + ReLU (contrastyou/arch/unet.py:22-23,25-26,40-41).

Checked here, through the C ABI:
  * the accumulated sums equal the f64 sum of the partial rows the same kernel writes in its other mode (exact to the
    2^-44 resolution of the split) -- for every producer kernel family and the first-layer kernels;
  * coefficients / outputs / gradients of the folding consumers against the finalize-launch path (which the oracle
    tests pin) and against torch's CPU batch norm;
  * run-to-run bit equality (integer adds commute: the order of arrival cannot matter);
  * a non-finite activation poisons the sums (NaN coefficients), as the f32 sums of the other path would;
  * a whole ConvChainFn block and a small U-Net step in both modes (CY_BN_ACC=0 / 1) agree to accumulation order.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from cyhip import ops
    return ops


def nhwc(t, dtype=None):
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def acc_sums(acc):
    """(sum, sum of squares) per channel as f64 from an accumulator [R][4][C]"""
    R, C = acc.R, acc.C
    w = acc.t[: R * 4 * C].view(R, 4, C).sum(0).double()
    assert int(acc.t[R * 4 * C:].sum()) == 0
    return w[0] / 4096.0 + w[1] / 2.0 ** 44, w[2] / 4096.0 + w[3] / 2.0 ** 44


# (N, H, W, C1, C2, Cout, mode, prologue, dtype): one case per producer kernel family
PRODUCERS = [
    (16, 56, 56, 128, 0, 128, 0, 0, torch.bfloat16),   # flow, 16 x 128 tiles
    (4, 112, 112, 64, 0, 64, 0, 0, torch.bfloat16),    # flow, 64 x 64 tiles (W % 16 == 0)
    (16, 14, 14, 512, 0, 512, 0, 0, torch.bfloat16),   # flow + split-K finish
    (16, 224, 224, 32, 0, 32, 0, 0, torch.bfloat16),   # streaming kernel
    (2, 56, 56, 32, 0, 64, 1, 0, torch.bfloat16),      # plane kernel (2x2 max on load)
    (2, 24, 24, 32, 0, 64, 0, 0, torch.float32),       # igemm kernel (f32)
    (2, 24, 24, 32, 0, 64, 0, 0, torch.float16),       # igemm kernel (f16, width not a multiple of 14 / 16)
]


@pytest.mark.parametrize("case", PRODUCERS, ids=lambda c: "x".join(str(v).replace("torch.", "") for v in c))
def test_accumulated_sums_equal_the_partial_rows(case):
    ops = _ops()
    N, H, W, C1, C2, Cout, mode, pro, dt = case
    g = torch.Generator().manual_seed(1)
    sh = 2 * H if mode == 1 else H
    x = nhwc(torch.randn(N, C1, sh, sh if mode == 1 else W, generator=g), dt)
    w = torch.randn(Cout, C1, 3, 3, generator=g).to(DEV) * 0.1
    wf, _ = ops.pack_weights(w, dt, want_dgrad=False)
    y0, part = ops.conv3x3_fwd(x, None, wf, Cout, mode=mode)
    y1, acc = ops.conv3x3_fwd(x, None, wf, Cout, mode=mode, stats_acc=True)
    assert torch.equal(y0, y1)
    s1, s2 = acc_sums(acc)
    p = part.double().sum(0)
    # the partial rows are f32; their exact sum is what the accumulator holds (each split is exact above 2^-21) -- but
    # for the streaming kernel, which sums its four wave rows in f32 before it adds (one add per workgroup)
    tol = 1e-6 if (C1 <= 64 and Cout <= 64 and H == 224) else 1e-9
    assert torch.allclose(s1, p[0], rtol=0, atol=tol * max(1.0, p[0].abs().max().item()))
    assert torch.allclose(s2, p[1], rtol=tol, atol=1e-9)
    # order independence: a second run gives the same words
    _, acc2 = ops.conv3x3_fwd(x, None, wf, Cout, mode=mode, stats_acc=True)
    assert torch.equal(acc.t, acc2.t)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_first_layer_accumulator(dt):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 1, 64, 64, generator=g).to(DEV)
    w = torch.randn(32, 1, 3, 3, generator=g).to(DEV)
    y0, part = ops.conv_first_fwd(x, w, dt)
    y1, acc = ops.conv_first_fwd(x, w, dt, stats_acc=True)
    assert torch.equal(y0, y1)
    s1, s2 = acc_sums(acc)
    p = part.double().sum(0)
    assert torch.allclose(s1, p[0], rtol=0, atol=1e-9 * max(1.0, p[0].abs().max().item()))
    assert torch.allclose(s2, p[1], rtol=1e-12, atol=1e-9)


def _chain_inputs(N, H, C, dt, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = nhwc(torch.randn(N, C, H, H, generator=g), dt)
    wa = (torch.randn(C, C, 3, 3, generator=g) * 0.08).to(DEV)
    wb = (torch.randn(C, C, 3, 3, generator=g) * 0.08).to(DEV)
    gm = (torch.rand(C, generator=g) + 0.5).to(DEV)
    bt = (torch.rand(C, generator=g) - 0.5).to(DEV)
    return x, wa, wb, gm, bt


# geometries whose second conv runs on: the flow kernel (fold in the kernel), the streaming kernel (fold in the kernel),
# the igemm kernel (fold launch, then coefficients from memory)
FOLD_CASES = [(8, 56, 128, torch.bfloat16), (16, 224, 32, torch.bfloat16), (2, 24, 64, torch.float32),
              (16, 28, 256, torch.float16)]


@pytest.mark.parametrize("case", FOLD_CASES, ids=lambda c: "x".join(str(v).replace("torch.", "") for v in c))
def test_folding_consumers_match_the_finalize_path(case):
    ops = _ops()
    N, H, C, dt = case
    x, wa, wb, gm, bt = _chain_inputs(N, H, C, dt)
    wfa, _ = ops.pack_weights(wa, dt, want_dgrad=False)
    wfb, _ = ops.pack_weights(wb, dt, want_dgrad=False)
    count = N * H * H
    eps = 1e-5
    # finalize path
    ya, part = ops.conv3x3_fwd(x, None, wfa, C)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    sc, sh, mean, istd = ops.bn_finalize(part, count, gm, bt, rm, rv, 0.1, eps, True, True, C, DEV)
    yb, _ = ops.conv3x3_fwd(ya, None, wfb, C, scale=sc, shift=sh)
    out_ref = ops.bn_relu_apply(ya, sc, sh)
    # accumulator path: conv b folds, then an apply folds the same state again (same coefficients)
    ya2, acc = ops.conv3x3_fwd(x, None, wfa, C, stats_acc=True)
    assert torch.equal(ya, ya2)
    st = ops.BnState(acc, gm, bt, count, eps, DEV)
    yb2, _ = ops.conv3x3_fwd(ya2, None, wfb, C, fold=st, want_stats=False)
    coef = st.coef.clone()
    for row, ref in zip(coef[:4], (sc, sh, mean, istd)):
        assert torch.allclose(row, ref, rtol=2e-6, atol=1e-7), (row - ref).abs().max()
    st.coef.zero_()
    out2 = ops.bn_relu_apply_fold(ya2, st)
    assert torch.equal(st.coef, coef)  # every consumer derives the same numbers
    tol = 1e-5 if dt == torch.float32 else 1.6e-2
    assert (yb2.float() - yb.float()).abs().max() <= tol * yb.float().abs().max()
    assert (out2.float() - out_ref.float()).abs().max() <= tol * out_ref.float().abs().max()
    # the pooled form
    out3, pooled = ops.bn_relu_apply_pool_fold(ya2, st)
    assert torch.equal(out3, out2)
    assert torch.equal(pooled.float(), F.max_pool2d(out2.float(), 2))
    # running statistics: one batched launch from the coefficient block = the finalize kernel's update
    rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    ops.bn_running_update([(st.coef, rm2, rv2, 0.1)])
    assert torch.allclose(rm2, rm, rtol=1e-6, atol=1e-7) and torch.allclose(rv2, rv, rtol=1e-6, atol=1e-7)
    # torch's own batch norm on the same raw conv output
    ref = F.relu(F.batch_norm(ya.float().cpu(), None, None, gm.cpu(), bt.cpu(), True, 0.1, eps))
    assert (out2.float().cpu() - ref).abs().max() <= (2e-5 if dt == torch.float32 else 1.6e-2) * ref.abs().max()


@pytest.mark.parametrize("case", [(16, 56, 128, torch.bfloat16), (4, 28, 64, torch.float32), (16, 14, 512, torch.float16)],
                         ids=lambda c: "x".join(str(v).replace("torch.", "") for v in c))
def test_backward_on_accumulators_matches_the_finalize_path(case):
    ops = _ops()
    N, H, C, dt = case
    g = torch.Generator().manual_seed(5)
    y = nhwc(torch.randn(N, C, H, H, generator=g), dt)
    da = nhwc(torch.randn(N, C, H, H, generator=g), dt)
    gm = (torch.rand(C, generator=g) + 0.5).to(DEV)
    bt = (torch.rand(C, generator=g) - 0.5).to(DEV)
    yf = y.float()
    mean = yf.mean((0, 2, 3))
    var = yf.var((0, 2, 3), unbiased=False)
    istd = (var + 1e-5).rsqrt()
    coef = torch.stack([gm * istd, bt - mean * gm * istd, mean, istd, var]).contiguous()
    dy0, dg0, db0 = ops.bn_relu_bwd(da, y, coef[0], coef[1], coef[2], coef[3], True)
    dy1, dg1, db1 = ops.bn_relu_bwd_acc(da, y, coef[0], True)
    tol = 1e-5 if dt == torch.float32 else 1.6e-2
    assert (dy1.float() - dy0.float()).abs().max() <= tol * dy0.float().abs().max()
    assert torch.allclose(dg1, dg0, rtol=2e-5, atol=2e-5 * dg0.abs().max().item())
    assert torch.allclose(db1, db0, rtol=2e-5, atol=2e-5 * db0.abs().max().item())
    # accumulation into live gradient buffers, and bit-equal repeats
    sink_g, sink_b = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    dy2, _, _ = ops.bn_relu_bwd_acc(da, y, coef[0], True, dgamma_out=sink_g, dbeta_out=sink_b)
    assert torch.equal(dy2, dy1)
    assert torch.equal(sink_g, 1 + dg1) and torch.equal(sink_b, 1 + db1)
    # the pool backward fills the accumulator itself
    if dt != torch.float32 or True:
        out = F.relu(yf * coef[0].view(1, -1, 1, 1) + coef[1].view(1, -1, 1, 1)).to(dt).contiguous(memory_format=torch.channels_last)
        dpool = nhwc(torch.randn(N, C, H // 2, H // 2, generator=g), dt)
        dx0, part = ops.maxpool2_bwd_bn(out, dpool, None, y, coef[0], coef[1], coef[2], coef[3])
        dx1, acc = ops.maxpool2_bwd_bn_acc(out, dpool, None, y, coef[0])
        assert torch.equal(dx0, dx1)
        if part is not None:
            s1, s2 = acc_sums(acc)
            p = part.double().sum(0)
            assert torch.allclose(s1, p[0], rtol=0, atol=1e-9 * max(1.0, p[0].abs().max().item()))
            assert torch.allclose(s2, p[1], rtol=0, atol=1e-9 * max(1.0, p[1].abs().max().item()))
            e0, _, _ = ops.bn_relu_bwd(dx0, y, coef[0], coef[1], coef[2], coef[3], True, partials=part)
            e1, _, _ = ops.bn_relu_bwd_acc(dx1, y, coef[0], True, acc=acc, acc_filled=True)
            assert (e1.float() - e0.float()).abs().max() <= tol * e0.float().abs().max()


def test_non_finite_activation_poisons_the_sums():
    ops = _ops()
    dt = torch.bfloat16
    x, wa, _, gm, bt = _chain_inputs(4, 56, 128, dt)
    x[1, 3, 5, 7] = float("inf")
    wfa, _ = ops.pack_weights(wa, dt, want_dgrad=False)
    ya, acc = ops.conv3x3_fwd(x, None, wfa, 128, stats_acc=True)
    st = ops.BnState(acc, gm, bt, 4 * 56 * 56, 1e-5, DEV)
    ops.bn_fold_coef(st)
    assert not torch.isfinite(st.coef[:2]).any()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_unet_step_agrees_between_the_two_paths(dt):
    """a small U-Net forward + backward with the accumulators on and off: same function, accumulation order apart"""
    ops = _ops()
    from contrastyou.arch.unet import UNet
    res = {}
    for mode in (True, False):
        ops.BN_ACC = mode
        try:
            torch.manual_seed(0)
            net = UNet(input_dim=1, num_classes=4, max_channel=128).to(DEV)
            net.compute_dtype = dt
            g = torch.Generator().manual_seed(9)
            x = torch.rand(3, 1, 64, 64, generator=g).to(DEV)
            for _ in range(2):  # (the second pass draws its accumulators from the arena)
                net.zero_grad()
                logits = net(x)
                logits.float().square().mean().backward()
            torch.cuda.synchronize()
            res[mode] = (logits.float().clone(), {n: p.grad.clone() for n, p in net.named_parameters()},
                         {n: b.clone() for n, b in net.named_buffers()})
        finally:
            ops.BN_ACC = True
    la, ga, ba = res[True]
    lb, gb, bb = res[False]
    tol = 2e-5 if dt == torch.float32 else 3e-2
    assert (la - lb).abs().max() <= tol * lb.abs().max()
    for n in gb:
        assert (ga[n] - gb[n]).abs().max() <= (1e-4 if dt == torch.float32 else 0.15) * gb[n].abs().max() + 1e-7, n
    for n in bb:
        assert torch.allclose(ba[n].float(), bb[n].float(), rtol=1e-5, atol=1e-6), n


def test_unet_step_with_the_fused_data_gradient():
    """the opt-in fused data gradient inside ConvChainFn.backward: same gradients as the two-launch form"""
    ops = _ops()
    from contrastyou.arch.unet import UNet
    res = {}
    for mode in (True, False):
        ops.DGRAD_BN = mode
        try:
            torch.manual_seed(0)
            net = UNet(input_dim=1, num_classes=4, max_channel=256).to(DEV)
            net.compute_dtype = torch.bfloat16
            g = torch.Generator().manual_seed(9)
            x = torch.rand(4, 1, 64, 64, generator=g).to(DEV)
            net.zero_grad()
            net(x).float().square().mean().backward()
            torch.cuda.synchronize()
            res[mode] = {n: p.grad.clone() for n, p in net.named_parameters()}
        finally:
            ops.DGRAD_BN = False
    for n, gb in res[False].items():
        assert (res[True][n] - gb).abs().max() <= 0.1 * gb.abs().max() + 1e-7, n


# (N, H, W, C = channels of dA / y, Cin = channels of the data gradient, split, dtype): the flow kernel's tilings with the
# backward prologue -- 16 x 128, 32 x 128, 64 x 64 on the 18-wide halo pitch, 16 x 64 four-wave tiles, 14-wide tiles,
# split outputs (the data gradient of a concat conv), split-K
DGRAD_BN_CASES = [
    (16, 56, 56, 128, 128, None, torch.bfloat16),
    (32, 56, 56, 128, 128, None, torch.bfloat16),
    (16, 112, 112, 64, 64, None, torch.bfloat16),
    (16, 28, 28, 256, 256, None, torch.bfloat16),
    (16, 28, 28, 256, 512, 256, torch.bfloat16),
    (16, 14, 14, 512, 512, None, torch.bfloat16),
    (16, 14, 14, 512, 256, None, torch.float16),
    (4, 64, 64, 64, 128, 64, torch.float16),
    (16, 56, 56, 128, 64, None, torch.bfloat16),
]


@pytest.mark.parametrize("case", DGRAD_BN_CASES, ids=lambda c: "x".join(str(v).replace("torch.", "") for v in c))
def test_dgrad_with_bn_backward_prologue(case):
    """cy_conv3x3_dgrad_bn = cy_bn_relu_bwd_apply_fold + cy_conv3x3_fwd: dy bit-equal (same arithmetic per element), the
    data gradient equal to accumulation order, dgamma / dbeta equal; twice the same bits"""
    ops = _ops()
    N, H, W, C, Cin, split, dt = case
    g = torch.Generator().manual_seed(11)
    y = nhwc(torch.randn(N, C, H, W, generator=g), dt)
    da = nhwc(torch.randn(N, C, H, W, generator=g), dt)
    gm = (torch.rand(C, generator=g) + 0.5).to(DEV)
    bt = (torch.rand(C, generator=g) - 0.5).to(DEV)
    w = (torch.randn(C, Cin, 3, 3, generator=g) * 0.05).to(DEV)   # forward weight [Cout = C][Cin]
    _, wd = ops.pack_weights(w, dt, want_dgrad=True)
    yf = y.float()
    mean = yf.mean((0, 2, 3))
    var = yf.var((0, 2, 3), unbiased=False)
    istd = (var + 1e-5).rsqrt()
    coef = torch.stack([gm * istd, bt - mean * gm * istd, mean, istd, var]).contiguous()
    ops.DGRAD_BN = True  # (opt-in in the product path: CY_DGRAD_BN=1)
    try:
        ok = ops.conv3x3_dgrad_bn_ok(da, Cin, split)
    finally:
        ops.DGRAD_BN = False
    assert ok, "every case here has a flow-kernel plan with room for the y buffer"
    acc = ops.bn_bwd_acc_new(N, C, H, W, False, DEV)
    ops.bn_bwd_reduce_acc(da, y, coef[0], acc)
    dy0, dg0, db0 = ops.bn_relu_bwd_acc(da, y, coef[0], True, acc=acc, acc_filled=True)
    ref, _ = ops.conv3x3_fwd(dy0, None, wd, Cin, want_stats=False, split=split)
    dx, dy1, dg1, db1 = ops.conv3x3_dgrad_bn(da, y, coef[0], acc, True, wd, Cin, split=split)
    torch.cuda.synchronize()
    assert torch.equal(dy1, dy0)
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0)
    refs = ref if split else (ref,)
    outs = dx if split else (dx,)
    for a, b in zip(outs, refs):
        assert (a.float() - b.float()).abs().max() <= 1.6e-2 * b.float().abs().max()
    dx2, dy2, _, _ = ops.conv3x3_dgrad_bn(da, y, coef[0], acc, True, wd, Cin, split=split)
    for a, b in zip(dx2 if split else (dx2,), outs):
        assert torch.equal(a, b)
    assert torch.equal(dy2, dy1)
    # against torch's own autograd of conv(relu(bn(y))) w.r.t. the conv input, f32 on the CPU
    yc = y.float().cpu().requires_grad_(True)
    a_ = F.relu(F.batch_norm(yc, None, None, gm.cpu(), bt.cpu(), True, 0.1, 1e-5))
    a_.backward(da.float().cpu())
    dxr = F.conv_transpose2d(yc.grad.to(dt).float(), w.cpu().to(dt).float(), padding=1)
    got = torch.cat([o.float().cpu() for o in outs], 1)
    assert (got - dxr).abs().max() <= 2.5e-2 * dxr.abs().max()


def test_upsample_backward_adds_the_consumers_bn_sums():
    """cy_upsample2_bwd_bn_acc: same gradient as cy_upsample2_bwd, and the accumulator holds what the reduce launch adds"""
    ops = _ops()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(21)
    N, C, H = 4, 128, 28
    dup = nhwc(torch.randn(N, C, 2 * H, 2 * H, generator=g), dt)
    y = nhwc(torch.randn(N, C, H, H, generator=g), dt)
    coef = torch.rand(5, C, generator=g).to(DEV) + 0.25
    coef[1] -= 0.75
    ref = ops.upsample2_bwd(dup)
    acc = ops.bn_bwd_acc_new(N, C, H, H, False, DEV)
    dx = ops.upsample2_bwd_bn_acc(dup, y, coef[0], acc)
    assert torch.equal(dx, ref)
    acc2 = ops.bn_bwd_acc_new(N, C, H, H, False, DEV)
    ops.bn_bwd_reduce_acc(ref, y, coef[0], acc2)
    a1, a2 = acc_sums(acc)
    b1, b2 = acc_sums(acc2)
    assert torch.allclose(a1, b1, rtol=0, atol=2e-6 * b1.abs().max().item())
    assert torch.allclose(a2, b2, rtol=0, atol=2e-6 * b2.abs().max().item())


def test_producer_side_sums_are_dropped_when_other_gradients_join():
    """a block output with TWO consumers (the decoder's upsample and a loss on the features): autograd sums their
    gradients, so the sums the upsample backward added describe only its own share and must not be used"""
    ops = _ops()
    from contrastyou.arch.unet import UNet
    res = {}
    for mode in (True, False):
        ops.BN_ACC = mode
        try:
            torch.manual_seed(0)
            net = UNet(input_dim=1, num_classes=4, max_channel=128).to(DEV)
            net.compute_dtype = torch.float32
            feats = {}
            h = net.get_module("Up_conv4").register_forward_hook(lambda m, i, o: feats.__setitem__("f", o))
            g = torch.Generator().manual_seed(9)
            x = torch.rand(3, 1, 64, 64, generator=g).to(DEV)
            net.zero_grad()
            logits = net(x)
            (logits.float().square().mean() + feats["f"].float().square().mean()).backward()
            h.remove()
            torch.cuda.synchronize()
            res[mode] = {n: p.grad.clone() for n, p in net.named_parameters()}
        finally:
            ops.BN_ACC = True
    for n, gb in res[False].items():
        assert (res[True][n] - gb).abs().max() <= 1e-4 * gb.abs().max() + 1e-8, n


# (N, H, W, C of dy, Cin of the data gradient, split, dtype)
DGRAD_DZ_CASES = [
    (16, 56, 56, 128, 128, None, torch.bfloat16),
    (32, 56, 56, 128, 128, None, torch.bfloat16),
    (16, 112, 112, 64, 64, None, torch.bfloat16),
    (16, 28, 28, 256, 256, None, torch.bfloat16),
    (16, 28, 28, 256, 512, 256, torch.bfloat16),
    (6, 28, 28, 128, 128, None, torch.float16),
    (3, 40, 48, 64, 128, 64, torch.float16),
]


@pytest.mark.parametrize("case", DGRAD_DZ_CASES, ids=lambda c: "x".join(str(v).replace("torch.", "") for v in c))
def test_dgrad_epilogue_adds_the_backward_sums(case):
    """cy_conv3x3_dgrad_dz: the same data gradient as cy_conv3x3_fwd, and the accumulator holds what the reduce launch
    over (its output, y) adds -- for the whole output, or the second part of a split output"""
    ops = _ops()
    N, H, W, C, Cin, split, dt = case
    g = torch.Generator().manual_seed(31)
    dy = nhwc(torch.randn(N, C, H, W, generator=g), dt)
    w = (torch.randn(C, Cin, 3, 3, generator=g) * 0.05).to(DEV)
    _, wd = ops.pack_weights(w, dt, want_dgrad=True)
    Cs = Cin - split if split else Cin
    y = nhwc(torch.randn(N, Cs, H, W, generator=g), dt)
    coef = torch.rand(5, Cs, generator=g).to(DEV) + 0.25
    coef[1] -= 0.75
    ops.DGRAD_DZ = True  # (opt-in in the product path: CY_DGRAD_DZ=1)
    try:
        ok = ops.conv3x3_dgrad_dz_ok(dy, Cin, split, split or 0, Cs)
    finally:
        ops.DGRAD_DZ = False
    if not ok:
        pytest.skip("no 16-row flow-kernel plan without split-K for this geometry")
    ref, _ = ops.conv3x3_fwd(dy, None, wd, Cin, want_stats=False, split=split)
    acc = ops.bn_bwd_acc_new(N, Cs, H, W, False, DEV)
    out = ops.conv3x3_dgrad_dz(dy, wd, Cin, y, coef[0], acc, split=split)
    refs = ref if split else (ref,)
    outs = out if split else (out,)
    for a, b in zip(outs, refs):
        assert torch.equal(a, b)
    acc2 = ops.bn_bwd_acc_new(N, Cs, H, W, False, DEV)
    ops.bn_bwd_reduce_acc(refs[-1], y, coef[0], acc2)
    a1, a2 = acc_sums(acc)
    b1, b2 = acc_sums(acc2)
    assert torch.allclose(a1, b1, rtol=0, atol=3e-6 * b1.abs().max().item()), (a1 - b1).abs().max()
    assert torch.allclose(a2, b2, rtol=0, atol=3e-6 * b2.abs().max().item()), (a2 - b2).abs().max()
    acc3 = ops.bn_bwd_acc_new(N, Cs, H, W, False, DEV)
    ops.conv3x3_dgrad_dz(dy, wd, Cin, y, coef[0], acc3, split=split)
    assert torch.equal(acc3.t, acc.t)
