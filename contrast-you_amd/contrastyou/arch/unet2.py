"""`Block` of the reference's second backbone (contrastyou/arch/unet2.py:208-224):
Conv2d(3x3, padding 1, bias) -> GroupNorm(groups, C) -> SiLU, on the HIP kernels.

`proj` / `norm` / `act` are parameter holders with the reference's names (checkpoints
interchange); forward runs the bias-free implicit-GEMM convolution and folds the conv bias into
the GroupNorm+SiLU kernels (cyhip.functions.Conv3x3Fn / GNSiLUFn).

`UNet2` (contrastyou/arch/unet2.py:22-135), the second `get_arch` target: every ResnetBlock's two
3x3 conv + GroupNorm + SiLU stages -- 18 of them, where the network's FLOPs are -- run on those HIP
kernels, and so does the glue around them (round 3): the 7x7 stem, the 1x1 residual / qkv / output
projections, the 4x4 stride-2 down convolution and the 4x4 stride-2 transposed up convolution are
im2col + strided f32-MFMA GEMM + col2im, the channel LayerNorm, the linear attention and the bottleneck's
softmax attention are the kernels of csrc/cy_unet2.hip (cyhip.glue).  What torch still does here is
memory, views, `cat`, the residual adds and dtype casts at the block borders.  UNet2 is not on the
SemiSupervisedEpocher + InfoNCE hot path (it has no `until=` / `arch_elements`,
semi_seg/hooks/infonce.py:98 cannot attach to it): these kernels are sized for correctness and
full-chip launches, not tuned per shape.  Module and parameter names are the reference's
(checkpoints interchange).  `with_time_emb=True` (round 4): `SinusoidalPosEmb`, the two small MLPs
(`time_mlp`, every `ResnetBlock.mlp`) and the `x * (scale + 1) + shift` modulation between GroupNorm and
SiLU run on HIP kernels as well (cy_sinusoidal_emb, cy_linear_*, cy_act_*, cy_gn_silu_mod_*); like the
reference, `forward(x, time)` then needs a `time` tensor of shape [B].
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from cyhip import ops
from cyhip.functions import ActFn, Conv3x3Fn, GNSiLUFn, GNSiLUModFn, LinearFn, compute_dtype_for
from cyhip.glue import AttentionFn, ChanLayerNormFn, Conv2dFn, ConvTranspose2dFn, LinearAttentionFn

__all__ = ["Block", "UNet2"]


class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        if dim_out % 8 or dim_out % groups:
            raise NotImplementedError("the HIP block needs output channel counts that are multiples of 8 "
                                      f"(and of `groups`), got {dim} -> {dim_out}, groups={groups}")
        self.proj = nn.Conv2d(dim, dim_out, 3, padding=1)
        self.norm = nn.GroupNorm(groups, dim_out)
        self.act = nn.SiLU()
        self.compute_dtype: Optional[torch.dtype] = None

    def forward(self, x: Tensor, scale_shift=None) -> Tensor:
        ops.require_gpu(x)
        dt = compute_dtype_for(x, self.compute_dtype)
        w = self.proj.weight
        pad = (-x.shape[1]) % 8
        if pad:  # (UNet2's stem has 10 channels: zero channels change nothing, the kernels load 8 at a time)
            x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, pad))
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, pad))
        x = ops.to_nhwc(x if x.dtype == dt else x.to(dt))
        y = Conv3x3Fn.apply(x, w)
        if scale_shift is not None:  # (scale, shift) of the time embedding, [B, C] or [B, C, 1, 1] (unet2.py:216-220)
            scale, shift = (t.reshape(t.shape[0], -1) for t in scale_shift)
            return GNSiLUModFn.apply(y, self.proj.bias, self.norm.weight, self.norm.bias, scale, shift,
                                     self.norm.num_groups, self.norm.eps)
        return GNSiLUFn.apply(y, self.proj.bias, self.norm.weight, self.norm.bias, self.norm.num_groups,
                              self.norm.eps)


# ---- the rest of UNet2: parameter holders with the reference's names, forward on cyhip.glue ------------------
class _Conv2d(nn.Conv2d):
    """nn.Conv2d (square kernel / stride / padding, dilation 1, groups 1) whose forward is im2col + HIP GEMM"""

    def forward(self, x: Tensor) -> Tensor:  # noqa: D102
        assert self.dilation == (1, 1) and self.groups == 1 and self.stride[0] == self.stride[1] \
            and self.padding[0] == self.padding[1] and self.padding_mode == "zeros"
        return Conv2dFn.apply(x, self.weight, self.bias, self.stride[0], self.padding[0])


class _ConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(C, C, 4, 2, 1) of the reference's Upsample (unet2.py:176-177): HIP GEMM + col2im"""

    def forward(self, x: Tensor, output_size=None) -> Tensor:  # noqa: D102
        assert output_size is None and self.dilation == (1, 1) and self.groups == 1 and self.output_padding == (0, 0) \
            and self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1]
        return ConvTranspose2dFn.apply(x, self.weight, self.bias, self.stride[0], self.padding[0])


class _Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        y = self.fn(x)
        return y + x.to(y.dtype)


class _ChanLayerNorm(nn.Module):
    """normalisation over the channel axis of an NCHW map with [1,C,1,1] affine (unet2.py:183-194)"""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))

    def forward(self, x):
        return ChanLayerNormFn.apply(x, self.g, self.b, self.eps)


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = _ChanLayerNorm(dim)

    def forward(self, x):
        return self.fn(self.norm(x))


class _LinearAttention(nn.Module):
    """softmax(q over channels), softmax(k over positions), out = (k v^T)^T q (unet2.py:245-271)"""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale, self.heads, self.dim_head = dim_head ** -0.5, heads, dim_head
        hidden = dim_head * heads
        self.to_qkv = _Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(_Conv2d(hidden, dim, 1), _ChanLayerNorm(dim))

    def forward(self, x):
        return self.to_out(LinearAttentionFn.apply(self.to_qkv(x), self.heads, self.dim_head, self.scale))


class _Attention(nn.Module):
    """plain softmax attention over all positions of the bottleneck map (unet2.py:274-304)"""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale, self.heads, self.dim_head = dim_head ** -0.5, heads, dim_head
        hidden = dim_head * heads
        self.to_qkv = _Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = _Conv2d(hidden, dim, 1)

    def forward(self, x):
        return self.to_out(AttentionFn.apply(self.to_qkv(x), self.heads, self.dim_head, self.scale))


class _SinusoidalPosEmb(nn.Module):
    """[B] -> [B, dim]: sin / cos of time * exp(-i log(10000) / (dim/2 - 1)) (unet2.py:161-173)"""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t: Tensor) -> Tensor:
        return ops.sinusoidal_emb(t, self.dim)


def _linear(lin: nn.Linear, x: Tensor) -> Tensor:
    return LinearFn.apply(x, lin.weight, lin.bias, 0, 0.0)


class ResnetBlock(nn.Module):
    """Block -> Block + (1x1 conv | identity) shortcut (unet2.py:227-243); with `time_emb_dim` the block owns
    `mlp` = SiLU -> Linear(time_emb_dim, 2 dim_out), whose output modulates the first Block"""

    def __init__(self, dim, dim_out, *, time_emb_dim=None, groups=8):
        super().__init__()
        self.mlp = (nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2))
                    if time_emb_dim is not None else None)  # (parameter holders: forward runs cy_act_* / cy_linear_*)
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.res_conv = _Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def forward(self, x, time_emb=None):
        scale_shift = None
        if self.mlp is not None and time_emb is not None:
            e = _linear(self.mlp[1], ActFn.apply(time_emb, ops.ACT_SILU))
            scale_shift = e.chunk(2, dim=1)
        h = self.block2(self.block1(x, scale_shift=scale_shift))
        return h + self.res_conv(x).to(h.dtype)


class UNet2(nn.Module):
    def __init__(self, init_dim=None, num_classes=None, dim_mults=(1, 2, 4, 8), input_dim=3, dim=16,
                 with_time_emb=False, resnet_block_groups=8, learned_variance=False, **kwargs):
        super().__init__()
        self.channels = input_dim
        init_dim = init_dim if init_dim is not None else dim // 3 * 2
        self.init_conv = _Conv2d(input_dim, init_dim, 7, padding=3)
        dims = [init_dim] + [dim * m for m in dim_mults]
        in_out = list(zip(dims[:-1], dims[1:]))
        if with_time_emb:  # (unet2.py:51-58)
            td = dim * 4
            self.time_mlp = nn.Sequential(_SinusoidalPosEmb(dim), nn.Linear(dim, td), nn.GELU(), nn.Linear(td, td))
        else:
            td = None
            self.time_mlp = None
        g = resnet_block_groups
        n_res = len(in_out)
        self.downs, self.ups = nn.ModuleList([]), nn.ModuleList([])
        for i, (cin, cout) in enumerate(in_out):
            last = i >= n_res - 1
            self.downs.append(nn.ModuleList([
                ResnetBlock(cin, cout, time_emb_dim=td, groups=g), ResnetBlock(cout, cout, time_emb_dim=td, groups=g),
                _Residual(_PreNorm(cout, _LinearAttention(cout))),
                nn.Identity() if last else _Conv2d(cout, cout, 4, 2, 1)]))
        mid = dims[-1]
        self.mid_block1 = ResnetBlock(mid, mid, time_emb_dim=td, groups=g)
        self.mid_attn = _Residual(_PreNorm(mid, _Attention(mid)))
        self.mid_block2 = ResnetBlock(mid, mid, time_emb_dim=td, groups=g)
        for i, (cin, cout) in enumerate(reversed(in_out[1:])):
            last = i >= n_res - 1
            self.ups.append(nn.ModuleList([
                ResnetBlock(cout * 2, cin, time_emb_dim=td, groups=g), ResnetBlock(cin, cin, time_emb_dim=td, groups=g),
                _Residual(_PreNorm(cin, _LinearAttention(cin))),
                nn.Identity() if last else _ConvTranspose2d(cin, cin, 4, 2, 1)]))
        self.out_dim = num_classes if num_classes is not None else input_dim * (2 if learned_variance else 1)
        self.num_classes = num_classes
        self.final_conv = nn.Sequential(ResnetBlock(dim, dim, groups=g), _Conv2d(dim, self.out_dim, 1))
        self._compute_dtype: Optional[torch.dtype] = None

    @property
    def compute_dtype(self):
        return self._compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dt):
        self._compute_dtype = dt
        for m in self.modules():
            if isinstance(m, Block):
                m.compute_dtype = dt

    def forward(self, x, time=None):
        ops.require_gpu(x)
        t = None
        if self.time_mlp is not None:
            if time is None:  # (the reference fails inside SinusoidalPosEmb here: unet2.py:104,166-171)
                raise TypeError("UNet2(with_time_emb=True).forward needs `time` ([B] tensor)")
            ops.require_gpu(time)
            m = self.time_mlp
            t = _linear(m[3], ActFn.apply(_linear(m[1], m[0](time)), ops.ACT_GELU))
        return self._forward(x, t)  # (the glue computes in f32 whatever the blocks' storage type is)

    def _forward(self, x, t=None):
        x = self.init_conv(x)
        skips = []
        for block1, block2, attn, down in self.downs:
            x = attn(block2(block1(x, t), t))
            skips.append(x)
            x = down(x)
        x = self.mid_block2(self.mid_attn(self.mid_block1(x, t)), t)
        for block1, block2, attn, up in self.ups:
            x = up(attn(block2(block1(torch.cat((x, skips.pop()), dim=1), t), t)))
        return self.final_conv(x)  # (final_conv's ResnetBlock has no time MLP: unet2.py:93-95)

    def switch_grad(self, **kwargs):
        from contextlib import nullcontext
        return nullcontext()

    def switch_bn_track(self, **kwargs):
        from contextlib import nullcontext
        return nullcontext()
