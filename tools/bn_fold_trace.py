import sys
from pathlib import Path
import torch
REPO = Path("/root/repo")
sys.path.insert(0, str(REPO / "contrast-you_amd"))
from cyhip import ops
dev, dt = "cuda", torch.bfloat16
for N, C, H in ((16, 32, 224), (16, 64, 112), (16, 128, 56), (16, 256, 28), (16, 512, 14)):
    y = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
    da = ops.empty_nhwc(N, C, H, H, dt, dev).normal_()
    gm, bt = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
    cnt = N * H * H
    coef = torch.rand(5, C, device=dev) + 0.5
    for R in (1, 4, 32 if C <= 64 else (8 if C <= 256 else 4)):
        t = torch.zeros(R * 4 * C + R, dtype=torch.int64, device=dev)
        acc = ops.BnAccBuf(t, R, C)
        st = ops.BnState(acc, gm, bt, cnt, 1e-5, dev)
        torch.cuda.synchronize()
        for _ in range(3):
            ops.bn_relu_apply_fold(y, st)
            torch.cuda.synchronize()
        for _ in range(3):
            ops.bn_relu_bwd_acc(da, y, coef[0], True, acc=acc, acc_filled=True)
            torch.cuda.synchronize()
    for _ in range(3):
        ops.bn_relu_apply(y, coef[0], coef[1])
        torch.cuda.synchronize()
    kc = torch.zeros(2 * C, device=dev)
    dy = torch.empty_like(y)
    for _ in range(3):
        ops._lib.call("cy_bn_relu_bwd_apply", da.data_ptr(), C, y.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), kc.data_ptr(), dy.data_ptr(), cnt, C, ops.dtype_code(dt), ops._stream())
        torch.cuda.synchronize()
