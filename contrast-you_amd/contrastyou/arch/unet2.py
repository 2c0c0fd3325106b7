"""`Block` of the reference's second backbone (contrastyou/arch/unet2.py:208-224):
Conv2d(3x3, padding 1, bias) -> GroupNorm(groups, C) -> SiLU, on the HIP kernels.

`proj` / `norm` / `act` are parameter holders with the reference's names (checkpoints
interchange); forward runs the bias-free implicit-GEMM convolution and folds the conv bias into
the GroupNorm+SiLU kernels (cyhip.functions.Conv3x3Fn / GNSiLUFn).  The rest of UNet2 (linear
attention, 4x4 strided / transposed convolutions, time embeddings) is outside this build's scope
("not reachable from the InfoNCE hook"); `get_arch("unet2")` says so.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from cyhip import ops
from cyhip.functions import Conv3x3Fn, GNSiLUFn, compute_dtype_for

__all__ = ["Block"]


class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        if dim % 8 or dim_out % 8 or dim_out % groups:
            raise NotImplementedError("the HIP block needs channel counts that are multiples of 8 "
                                      f"(and of `groups`), got {dim} -> {dim_out}, groups={groups}")
        self.proj = nn.Conv2d(dim, dim_out, 3, padding=1)
        self.norm = nn.GroupNorm(groups, dim_out)
        self.act = nn.SiLU()
        self.compute_dtype: Optional[torch.dtype] = None

    def forward(self, x: Tensor, scale_shift=None) -> Tensor:
        if scale_shift is not None:
            raise NotImplementedError("time-embedding scale/shift is not used by the segmentation path")
        ops.require_gpu(x)
        dt = compute_dtype_for(x, self.compute_dtype)
        x = ops.to_nhwc(x if x.dtype == dt else x.to(dt))
        y = Conv3x3Fn.apply(x, self.proj.weight)
        return GNSiLUFn.apply(y, self.proj.bias, self.norm.weight, self.norm.bias, self.norm.num_groups,
                              self.norm.eps)
