"""Helper of tests/test_gpu_switches.py (not a test): evaluates two C2-shaped layers and a small U-Net step through the
HIP path under whatever CY_* switches the environment carries (the library reads them once per process) and saves the
results.   python tests/switch_case.py <out.pt>"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
for p in (REPO, REPO / "contrast-you_amd"):
    sys.path.insert(0, str(p))
from cyhip import ops  # noqa: E402

DEV, DT = "cuda", torch.bfloat16


def nhwc(t):
    return t.to(DT).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def layer(tag, N, H, Cin, Cout, out):
    g = torch.Generator().manual_seed(len(tag) * 7 + N)
    x = nhwc(torch.randn(N, Cin, H, H, generator=g))
    dy = nhwc(torch.randn(N, Cout, H, H, generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.06).to(DEV)
    sc = (torch.rand(Cin, generator=g) + 0.5).to(DEV)
    sh = (torch.rand(Cin, generator=g) - 0.5).to(DEV)
    wf, wd = ops.pack_weights(w, DT)
    y, part = ops.conv3x3_fwd(x, None, wf, Cout, scale=sc, shift=sh)
    out[f"{tag}_fwd"] = y.float().cpu()
    out[f"{tag}_stats"] = part.double().sum(0).float().cpu()
    dx, _ = ops.conv3x3_fwd(dy, None, wd, Cin, want_stats=False)
    out[f"{tag}_dgrad"] = dx.float().cpu()
    out[f"{tag}_wgrad"] = ops.conv3x3_wgrad(x, None, dy, scale=sc, shift=sh).cpu()
    sink = torch.zeros(Cout, Cin, 3, 3, device=DEV)
    ops.conv3x3_wgrad_pair(x, None, dy, sc, sh, x, None, dy, sc, sh, mode=0, out=sink)
    out[f"{tag}_wgrad_pair"] = sink.cpu()


def unet_step(out):
    from contrastyou.arch.unet import UNet
    from contrastyou.optim.fused_radam import FusedRAdam
    torch.manual_seed(0)
    net = UNet(input_dim=1, num_classes=4, max_channel=128).to(DEV)
    net.compute_dtype = DT
    opt = FusedRAdam(net.parameters(), lr=1e-3)
    opt.zero_grad()
    g = torch.Generator().manual_seed(9)
    xa, xb = torch.rand(2, 1, 64, 64, generator=g).to(DEV), torch.rand(3, 1, 64, 64, generator=g).to(DEV)
    la = net(xa)                       # two passes into the same gradients: the paired weight gradients' case
    lb = net(xb, until="Conv5")
    (la.float().square().mean() + lb.float().square().mean()).backward()
    torch.cuda.synchronize()
    out["unet_logits"] = la.float().cpu()
    for n, p in net.named_parameters():
        out[f"unet_grad_{n}"] = p.grad.detach().float().cpu().clone()
    for n, b in net.named_buffers():
        out[f"unet_buf_{n}"] = b.detach().float().cpu().clone()


def first_layer(out):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 1, 64, 64, generator=g).to(DEV)
    w = (torch.randn(32, 1, 3, 3, generator=g) * 0.3).to(DEV)
    y, part = ops.conv_first_fwd(x, w, DT)
    out["first_fwd"] = y.float().cpu()
    out["first_stats"] = part.double().sum(0).float().cpu()
    dy = nhwc(torch.randn(2, 32, 64, 64, generator=g))
    out["first_wgrad"] = ops.conv_first_wgrad(x, dy).cpu()


if __name__ == "__main__":
    res = {}
    first_layer(res)
    layer("conv3b", 4, 56, 128, 128, res)   # flow kernel, wave-specialised weight gradient
    layer("conv1b", 2, 224, 32, 32, res)    # streaming kernel, 32-channel weight-gradient blocks
    unet_step(res)
    torch.save(res, sys.argv[1])
