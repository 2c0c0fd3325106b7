"""2-D U-Net of the reference (contrastyou/arch/unet.py:16-246) on hand-written gfx950 kernels.

Same constructor, attribute names, parameter/buffer names (checkpoints interchange:
`_Conv1.conv.0.weight`, `_Conv1.conv.1.running_mean`, ..., `_Up5.up.1.weight`,
`_Deconv_1x1.bias`), `forward(x, until=)`, `get_channel_dim`, `get_module`, `switch_grad`,
`switch_bn_track` as the reference class.  What differs is the execution: every named block
is ONE autograd Function (cyhip.functions.ConvChainFn) over HIP kernels, activations are
NHWC (channels_last) bf16/f32, and

  * nn.MaxPool2d(2)      -> second output of the block's last BN+ReLU launch (cy_bn_relu_apply_pool),
  * nn.Upsample(x2)      -> CY_SRC_UP2 load mode of the _UpConv conv,
  * torch.cat((skip,up)) -> two-source load of the decoder block's first conv,
  * BN+ReLU after conv 1 -> prologue of conv 2 (only raw conv outputs hit HBM inside a block).

The nn.Conv2d / nn.BatchNorm2d children are parameter and running-statistic holders only;
their own forward() is never called.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

from collections import OrderedDict
from contextlib import contextmanager
from enum import Enum
from functools import partial
from typing import Optional

import torch
from torch import Tensor, nn

from cyhip import ops
from cyhip.functions import (ChainCfg, ConvChainFn, HeadFn, begin_pass, compute_dtype_for, defer_batch_counters,
                             prepack)

from ._base import _check_params, _complete_arch_start2end
from .utils import get_bn_track, get_requires_grad

__all__ = ["UNet", "UNetFeatureMapEnum"]


def _conv3x3(cin: int, cout: int) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False)


class _ConvBlock(nn.Module):
    """[conv3x3 -> BN -> ReLU] x 2   (reference unet.py:16-31)"""

    def __init__(self, in_ch: int, out_ch: int, momentum: float = 0.1):
        super().__init__()
        self.conv = nn.Sequential(
            _conv3x3(in_ch, out_ch), nn.BatchNorm2d(out_ch, momentum=momentum), nn.ReLU(inplace=True),
            _conv3x3(out_ch, out_ch), nn.BatchNorm2d(out_ch, momentum=momentum), nn.ReLU(inplace=True),
        )
        self._first = in_ch <= 4
        self._cfgs = {}
        self._pooled: Optional[Tensor] = None
        self.ready_tag: Optional[str] = None  # (UNet: "the gradients up to this block are final" mark)
        self.compute_dtype: Optional[torch.dtype] = None

    def _cfg(self, mode: int, pool_out: bool = False) -> ChainCfg:
        cfg = self._cfgs.get((mode, pool_out))
        if cfg is None:
            cfg = self._cfgs[(mode, pool_out)] = ChainCfg([self.conv[1], self.conv[4]], mode, self._first, pool_out)
        cfg.dtype = self.compute_dtype
        cfg.ready_tag = self.ready_tag
        return cfg

    def forward(self, x: Tensor, x2: Optional[Tensor] = None, *, pool: bool = False, pool_out: bool = False) -> Tensor:
        """pool: the input is max-pooled 2x2 on load.  pool_out: the launch that writes the block output also
        writes its 2x2 max (`take_pooled()`), so that the next block reads a plain tensor; the module's output
        stays the one tensor forward hooks / feature extractors expect."""
        c = self.conv
        mode = ops.CY_SRC_POOL2 if pool else ops.CY_SRC_DIRECT
        r = ConvChainFn.apply(self._cfg(mode, pool_out), x, x2, c[0].weight, c[1].weight, c[1].bias,
                              c[3].weight, c[4].weight, c[4].bias)
        if pool_out:
            r, self._pooled = r
        return r

    def take_pooled(self) -> Tensor:
        p, self._pooled = self._pooled, None
        return p


class _UpConv(nn.Module):
    """Upsample(x2, nearest) -> conv3x3 -> BN -> ReLU   (reference unet.py:34-46)"""

    def __init__(self, in_ch: int, out_ch: int, momentum: float = 0.1):
        super().__init__()
        self.up = nn.Sequential(
            nn.Upsample(scale_factor=2), _conv3x3(in_ch, out_ch),
            nn.BatchNorm2d(out_ch, momentum=momentum), nn.ReLU(inplace=True),
        )
        self._chain = ChainCfg([self.up[2]], ops.CY_SRC_UP2, False)
        self.compute_dtype: Optional[torch.dtype] = None
        self.ready_tag: Optional[str] = None

    def forward(self, x: Tensor) -> Tensor:
        u = self.up
        self._chain.dtype = self.compute_dtype
        self._chain.ready_tag = self.ready_tag
        return ConvChainFn.apply(self._chain, x, None, u[1].weight, u[2].weight, u[2].bias)


class _Head1x1(nn.Conv2d):
    """nn.Conv2d(C, num_classes, 1) whose forward is the HIP head kernel (f32 logits)."""

    def forward(self, x: Tensor) -> Tensor:  # noqa: D102
        return HeadFn.apply(x, self.weight, self.bias)


class UNet(nn.Module):
    layer_dimension = {"Conv1": 1, "Conv2": 2, "Conv3": 4, "Conv4": 8, "Conv5": 16, "Up_conv5": 8,
                       "Up_conv4": 4, "Up_conv3": 2, "Up_conv2": 1, "Deconv_1x1": None}
    encoder_names = ("Conv1", "Conv2", "Conv3", "Conv4", "Conv5")
    decoder_names = ("Up5", "Up_conv5", "Up4", "Up_conv4", "Up3", "Up_conv3", "Up2", "Up_conv2",
                     "Deconv_1x1")
    arch_elements = encoder_names + decoder_names

    def __init__(self, input_dim=3, num_classes=1, max_channel=256, momentum=0.1):
        super().__init__()
        assert max_channel % 16 == 0 and max_channel >= 128, max_channel
        if input_dim > 4:
            raise NotImplementedError("the HIP first-layer kernel handles input_dim <= 4")
        self._input_dim, self._num_classes, self._max_channel = input_dim, num_classes, max_channel
        ch = self.get_channel_dim

        for i in range(1, 5):  # kept for structural parity; pooling happens in the conv loads
            setattr(self, f"_max_pool{i}", nn.MaxPool2d(kernel_size=2, stride=2))

        self._Conv1 = _ConvBlock(input_dim, ch("Conv1"), momentum)
        self._Conv2 = _ConvBlock(ch("Conv1"), ch("Conv2"), momentum)
        self._Conv3 = _ConvBlock(ch("Conv2"), ch("Conv3"), momentum)
        self._Conv4 = _ConvBlock(ch("Conv3"), ch("Conv4"), momentum)
        self._Conv5 = _ConvBlock(ch("Conv4"), ch("Conv5"), momentum)

        self._Up5 = _UpConv(ch("Conv5"), ch("Up_conv5"), momentum)
        self._Up_conv5 = _ConvBlock(ch("Conv5"), ch("Up_conv5"), momentum)
        self._Up4 = _UpConv(ch("Up_conv5"), ch("Up_conv4"), momentum)
        self._Up_conv4 = _ConvBlock(ch("Up_conv5"), ch("Up_conv4"), momentum)
        self._Up3 = _UpConv(ch("Up_conv4"), ch("Up_conv3"), momentum)
        self._Up_conv3 = _ConvBlock(ch("Up_conv4"), ch("Up_conv3"), momentum)
        self._Up2 = _UpConv(ch("Up_conv3"), ch("Up_conv2"), momentum)
        self._Up_conv2 = _ConvBlock(ch("Up_conv3"), ch("Up_conv2"), momentum)

        self._Deconv_1x1 = _Head1x1(ch("Up_conv2"), num_classes, kernel_size=(1, 1), stride=(1, 1),
                                    padding=(0, 0))
        self._compute_dtype: Optional[torch.dtype] = None
        # Data parallel (contrastyou.optim.FusedRAdam): the backward pass is done with the decoder when Up5's
        # backward has run, with Conv5 / Conv4 after theirs -- 85 % of the parameters well before it ends.  Those
        # blocks leave a mark, their parameters carry its name: gradient buckets made of marked parameters are
        # all-reduced while the rest of the backward pass still runs.
        for tag, names in (("decoder", self.decoder_names), ("conv5", ("Conv5",)), ("conv4", ("Conv4",))):
            for n in names:
                for p in getattr(self, f"_{n}").parameters():
                    p.__dict__["_cy_ready_tag"] = tag
        self._Up5.ready_tag, self._Conv5.ready_tag, self._Conv4.ready_tag = "decoder", "conv5", "conv4"

    # ---- precision control (None: bf16 under autocast, else f32 verification mode) ----
    @property
    def compute_dtype(self) -> Optional[torch.dtype]:
        return self._compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dt: Optional[torch.dtype]):
        self._compute_dtype = dt
        for m in self.modules():
            if isinstance(m, (_ConvBlock, _UpConv)):
                m.compute_dtype = dt

    def forward(self, x: Tensor, until: str = None):
        if until and until not in self.layer_dimension:
            raise KeyError(f"`return_until` should be in {', '.join(self.layer_dimension.keys())},"
                           f" given {until}  ")
        ops.require_gpu(x)
        ops.note_home_stream(x.device)
        if x.dim() != 4 or x.shape[1] != self._input_dim:
            raise ValueError(f"expected [N,{self._input_dim},H,W], got {tuple(x.shape)}")
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError("spatial dims must be multiples of 16 (four 2x2 poolings)")
        # every 3x3 weight changes with the optimizer step: repack them all with one launch (a no-op
        # while the packed images are current, e.g. for the second pass of a step)
        begin_pass()
        prepack(self._packed_conv_weights(), compute_dtype_for(x, self._compute_dtype))
        with defer_batch_counters(x.device):
            return self._forward(x, until)

    def _packed_conv_weights(self):
        """the weights that go through the MFMA kernels (all 3x3 convolutions but the 1-4 channel stem)"""
        ws = []
        for m in self.modules():
            if isinstance(m, _ConvBlock):
                ws += [m.conv[3].weight] if m._first else [m.conv[0].weight, m.conv[3].weight]
            elif isinstance(m, _UpConv):
                ws.append(m.up[1].weight)
        return [w for w in ws if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()]

    def _forward(self, x: Tensor, until: Optional[str]):
        # nn.MaxPool2d(2) between the encoder blocks (reference unet.py:108-121): the pooled tensor is a second
        # output of the launch that writes the block output
        e1 = self._Conv1(x, pool_out=until != "Conv1")
        if until == "Conv1":
            return e1
        e2 = self._Conv2(self._Conv1.take_pooled(), pool_out=until != "Conv2")
        if until == "Conv2":
            return e2
        e3 = self._Conv3(self._Conv2.take_pooled(), pool_out=until != "Conv3")
        if until == "Conv3":
            return e3
        e4 = self._Conv4(self._Conv3.take_pooled(), pool_out=until != "Conv4")
        if until == "Conv4":
            return e4
        e5 = self._Conv5(self._Conv4.take_pooled())
        if until == "Conv5":
            return e5

        d5 = self._Up_conv5(e4, self._Up5(e5))  # cat((e4, up), dim=1) inside the conv loads
        if until == "Up_conv5":
            return d5
        d4 = self._Up_conv4(e3, self._Up4(d5))
        if until == "Up_conv4":
            return d4
        d3 = self._Up_conv3(e2, self._Up3(d4))
        if until == "Up_conv3":
            return d3
        d2 = self._Up_conv2(e1, self._Up2(d3))
        if until == "Up_conv2":
            return d2
        return self._Deconv_1x1(d2)

    def get_channel_dim(self, name: str) -> int:
        if name == "Deconv_1x1":
            return self._num_classes
        if name in self.layer_dimension:
            return int(self.layer_dimension[name] / 16 * self._max_channel)
        raise KeyError(name)

    def get_module(self, name: str) -> nn.Module:
        assert name in self.arch_elements, name
        return getattr(self, f"_{name}")

    @property
    def num_classes(self) -> int:
        return self._num_classes

    # ---- freeze / BN-tracking ranges (reference unet.py:193-242) ----
    def _range(self, start, end, include_start, include_end):
        _check_params(start, end, include_start, include_end, model=self)
        return _complete_arch_start2end(start or "Conv1", end or "Deconv_1x1", include_start=include_start,
                                        include_end=include_end, model=self)

    @contextmanager
    def switch_grad(self, enable=True, *, start: str = None, end: str = None, include_start=True,
                    include_end=True):
        names = self._range(start, end, include_start, include_end)
        prev = OrderedDict()
        for n in names:
            m = getattr(self, f"_{n}")
            prev[n] = get_requires_grad(m)
            m.requires_grad_(enable)
        try:
            yield self
        finally:
            for n, state in prev.items():
                getattr(self, f"_{n}").requires_grad_(state)

    @contextmanager
    def switch_bn_track(self, enable=True, *, start: str = None, end: str = None, include_start=True,
                        include_end=True):
        names = self._range(start, end, include_start, include_end)

        def _set(m, value):
            if hasattr(m, "track_running_stats"):
                m.track_running_stats = value

        prev = OrderedDict()
        for n in names:
            m = getattr(self, f"_{n}")
            try:
                prev[n] = get_bn_track(m)
            except RuntimeError:  # block without BN (Deconv_1x1)
                continue
            m.apply(partial(_set, value=enable))
        try:
            yield self
        finally:
            for n, state in prev.items():
                getattr(self, f"_{n}").apply(partial(_set, value=state))


class UNetFeatureMapEnum(Enum):
    Conv1 = "Conv1"
    Conv2 = "Conv2"
    Conv3 = "Conv3"
    Conv4 = "Conv4"
    Conv5 = "Conv5"
    Up_conv5 = "Up_conv5"
    Up_conv4 = "Up_conv4"
    Up_conv3 = "Up_conv3"
    Up_conv2 = "Up_conv2"
    Deconv_1x1 = "Deconv_1x1"
