"""The glue kernels of the second backbone (csrc/cy_unet2.hip, cyhip/glue.py) against torch's CPU operators on the
same inputs: strided batched GEMM (layouts, batches, split-K, bias, accumulate), K x K / 1 x 1 / transposed
convolutions (reference unet2.py:45,176-181), channel LayerNorm (:183-194), LinearAttention (:258-271) and Attention
(:289-302), forward and every gradient; f32, tolerance = accumulation order.  The end-to-end check against the
REFERENCE's own UNet2 outputs is tests/test_gpu_round2_rows.py::test_unet2_get_arch_matches_reference."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, rel, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs().max().item()
    assert err <= rel * (b.abs().max().item() + 1e-30), f"{what}: {err:.3e} vs max {b.abs().max().item():.3e}"


def test_gemm_strided_layouts_batches_splitk():
    from cyhip._lib import MatLayout
    from cyhip.glue import gemm
    g = torch.Generator().manual_seed(0)
    # plain, ragged sizes, bias, accumulate
    A, B, bias = torch.randn(70, 37, generator=g), torch.randn(37, 45, generator=g), torch.randn(45, generator=g)
    C0 = torch.randn(70, 45, generator=g)
    Cd = C0.to(DEV)
    gemm((A.to(DEV), 0), MatLayout(37, 1, 0, 0), (B.to(DEV), 0), MatLayout(45, 1, 0, 0), (Cd, 0), MatLayout(45, 1, 0, 0),
         70, 45, 37, bias=bias.to(DEV), alpha=0.5, accumulate=True)
    close(Cd, C0 + 0.5 * (A @ B) + bias, 1e-5, "gemm + bias + accumulate")
    # both operands transposed in memory, long K with split-K (run twice: bit-identical)
    At, Bt = torch.randn(5000, 33, generator=g), torch.randn(20, 5000, generator=g)
    outs = []
    for _ in range(2):
        Cd = torch.empty(33, 20, device=DEV)
        gemm((At.to(DEV), 0), MatLayout(1, 33, 0, 0), (Bt.to(DEV), 0), MatLayout(1, 5000, 0, 0), (Cd, 0),
             MatLayout(20, 1, 0, 0), 33, 20, 5000, ksplit=7)
        outs.append(Cd)
    close(outs[0], At.t() @ Bt.t(), 2e-5, "gemm A^T B^T split-K")
    assert torch.equal(outs[0], outs[1])
    # two-level batch addressing head blocks of a wider matrix, output into a column slice
    nb1, nb2, n, dh, ld = 3, 2, 50, 8, 40
    X = torch.randn(nb1 * n, ld, generator=g)
    Wm = torch.randn(nb1, nb2, dh, dh, generator=g)
    out = torch.zeros(nb1 * n, ld)
    od = out.to(DEV)
    gemm((X.to(DEV), 16), MatLayout(ld, 1, n * ld, dh), (Wm.to(DEV), 0), MatLayout(dh, 1, nb2 * dh * dh, dh * dh),
         (od, 4), MatLayout(ld, 1, n * ld, dh), n, dh, dh, nb1=nb1, nb2=nb2)
    ref = out.clone().view(nb1, n, ld)
    for b in range(nb1):
        for h in range(nb2):
            ref[b, :, 4 + h * dh: 4 + (h + 1) * dh] = X.view(nb1, n, ld)[b, :, 16 + h * dh: 16 + (h + 1) * dh] @ Wm[b, h]
    close(od, ref.view(-1, ld), 1e-5, "batched head blocks")


@pytest.mark.parametrize("cfg", [(1, 4, 7, 1, 3, 20, 24), (16, 16, 4, 2, 1, 12, 16), (5, 9, 1, 1, 0, 6, 10),
                                 (8, 8, 3, 2, 1, 9, 11)])
def test_conv2d_matches_torch(cfg):
    from cyhip.glue import Conv2dFn
    cin, cout, k, stride, pad, H, W = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    y_ref = F.conv2d(x, w, b, stride, pad)
    dy = torch.randn(y_ref.shape, generator=g)
    (y_ref * dy).sum().backward()
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = Conv2dFn.apply(xd, wd, bd, stride, pad)
    assert tuple(y.shape) == tuple(y_ref.shape)
    close(y, y_ref, 2e-5, "conv fwd")
    (y * dy.to(DEV)).sum().backward()
    close(xd.grad, x.grad, 3e-5, "conv dx")
    close(wd.grad, w.grad, 3e-5, "conv dw")
    close(bd.grad, b.grad, 3e-5, "conv db")


@pytest.mark.parametrize("cfg", [(16, 16, 4, 2, 1, 7, 9), (6, 10, 4, 2, 1, 5, 5)])
def test_conv_transpose2d_matches_torch(cfg):
    from cyhip.glue import ConvTranspose2dFn
    cin, cout, k, stride, pad, H, W = cfg
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(cin, cout, k, k, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(cout, generator=g).requires_grad_(True)
    y_ref = F.conv_transpose2d(x, w, b, stride, pad)
    dy = torch.randn(y_ref.shape, generator=g)
    (y_ref * dy).sum().backward()
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ConvTranspose2dFn.apply(xd, wd, bd, stride, pad)
    assert tuple(y.shape) == tuple(y_ref.shape)
    close(y, y_ref, 2e-5, "convT fwd")
    (y * dy.to(DEV)).sum().backward()
    close(xd.grad, x.grad, 3e-5, "convT dx")
    close(wd.grad, w.grad, 3e-5, "convT dw")
    close(bd.grad, b.grad, 3e-5, "convT db")


@pytest.mark.parametrize("C", [4, 16, 100])
def test_chan_layernorm_matches_reference_formula(C):
    from cyhip.glue import ChanLayerNormFn
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(2, C, 9, 7, generator=g) * 2 + 0.5).requires_grad_(True)
    gam = (torch.randn(1, C, 1, 1, generator=g) + 1).requires_grad_(True)
    bet = torch.randn(1, C, 1, 1, generator=g).requires_grad_(True)
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)  # (the reference's expression, unet2.py:191-194)
    y_ref = (x - torch.mean(x, dim=1, keepdim=True)) / (var + 1e-5).sqrt() * gam + bet
    dy = torch.randn(y_ref.shape, generator=g)
    (y_ref * dy).sum().backward()
    xd, gd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, gam, bet))
    y = ChanLayerNormFn.apply(xd, gd, bd, 1e-5)
    close(y, y_ref, 2e-5, "LN fwd")
    (y * dy.to(DEV)).sum().backward()
    close(xd.grad, x.grad, 5e-5, "LN dx")
    close(gd.grad, gam.grad, 5e-5, "LN dg")
    close(bd.grad, bet.grad, 5e-5, "LN db")


def _heads(t, heads):
    b, c, h, w = t.shape
    return t.reshape(b, heads, c // heads, h * w)


@pytest.mark.parametrize("hw", [(6, 5), (40, 36)])
def test_linear_attention_core(hw):
    from cyhip.glue import LinearAttentionFn
    heads, dh = 4, 32
    H, W = hw
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(2, 3 * heads * dh, H, W, generator=g, requires_grad=True)
    q, k, v = (_heads(t, heads) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * dh ** -0.5
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out_ref = torch.einsum("bhde,bhdn->bhen", context, q).reshape(2, heads * dh, H, W)
    dy = torch.randn(out_ref.shape, generator=g)
    (out_ref * dy).sum().backward()
    qd = qkv.detach().to(DEV).requires_grad_(True)
    out = LinearAttentionFn.apply(qd, heads, dh, dh ** -0.5)
    close(out, out_ref, 3e-5, "linear attention fwd")
    (out * dy.to(DEV)).sum().backward()
    close(qd.grad, qkv.grad, 1e-4, "linear attention dqkv")


@pytest.mark.parametrize("hw", [(4, 4), (9, 14)])
def test_softmax_attention_core(hw):
    from cyhip.glue import AttentionFn
    heads, dh = 4, 32
    H, W = hw
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(2, 3 * heads * dh, H, W, generator=g, requires_grad=True)
    q, k, v = (_heads(t, heads) for t in qkv.chunk(3, dim=1))
    sim = torch.einsum("bhdi,bhdj->bhij", q * dh ** -0.5, k)
    attn = (sim - sim.amax(dim=-1, keepdim=True).detach()).softmax(dim=-1)
    out_ref = torch.einsum("bhij,bhdj->bhid", attn, v).transpose(-1, -2).reshape(2, heads * dh, H, W)
    dy = torch.randn(out_ref.shape, generator=g)
    (out_ref * dy).sum().backward()
    qd = qkv.detach().to(DEV).requires_grad_(True)
    out = AttentionFn.apply(qd, heads, dh, dh ** -0.5)
    close(out, out_ref, 3e-5, "attention fwd")
    (out * dy.to(DEV)).sum().backward()
    close(qd.grad, qkv.grad, 1e-4, "attention dqkv")


def test_unet2_runs_on_the_hip_glue_only(monkeypatch):
    """no library convolution / matmul left in UNet2's forward + backward: the torch entry points the glue used to go
    through are poisoned for the duration of one step"""
    from contrastyou.arch import get_arch
    net = get_arch("unet2", input_dim=1, num_classes=4, dim=8).to(DEV)
    x = torch.randn(2, 1, 32, 32, device=DEV)

    def boom(*a, **k):
        raise AssertionError("library convolution / matmul reached from UNet2")

    for name in ("conv2d", "conv_transpose2d"):
        monkeypatch.setattr(F, name, boom)
    monkeypatch.setattr(torch, "matmul", boom)
    monkeypatch.setattr(torch, "einsum", boom)
    monkeypatch.setattr(torch.nn.Conv2d, "_conv_forward", boom)
    y = net(x)
    y.float().sum().backward()
    assert torch.isfinite(y).all() and all(p.grad is not None for p in net.parameters())
