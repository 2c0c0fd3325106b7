"""Development aid: the dense projector at the dense InfoNCE hook's size (32 x 224^2 x 32, 256 hidden units, 20 x 20 bins),
forward and backward, a few iterations each -- run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "contrast-you_amd"))
from cyhip import ops  # noqa: E402


def main():
    dev, dt = "cuda", torch.bfloat16
    n, c, hw, s, hid = 32, 32, 224, 20, 256
    x = torch.randn(n, hw, hw, c, device=dev).to(dt).permute(0, 3, 1, 2)
    w1 = torch.randn(hid, c, device=dev) * 0.1
    b1 = torch.randn(hid, device=dev) * 0.1
    g = torch.randn(n * s * s, hid, device=dev)
    for _ in range(int(os.environ.get("ITERS", 10))):
        ops.dense_proj_fwd(x, w1, b1, (s, s), None)
        ops.dense_proj_bwd(x, w1, b1, (s, s), None, g, True, True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
