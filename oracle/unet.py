"""ORACLE (test infrastructure, not product code): CPU restatement of the reference U-Net.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
It restates, with stock torch CPU ops in f32/f64, the arithmetic of
    contrastyou/arch/unet.py:16-31   _ConvBlock  = [conv3x3(no bias) -> BN2d -> ReLU] x 2
    contrastyou/arch/unet.py:34-46   _UpConv     = Upsample(x2 nearest) -> conv3x3 -> BN2d -> ReLU
    contrastyou/arch/unet.py:105-177 UNet.forward: 4 x MaxPool2d(2), cat((skip, up), dim=1), 1x1 head
as a pure function of a reference-named state dict.  Pinned against the reference itself:
tests/golden/unet_small.npz was produced by importing /root/reference (gen_goldens.py) and
tests/test_oracle_golden.py checks this file against it.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

LAYER_DIMENSION = {"Conv1": 1, "Conv2": 2, "Conv3": 4, "Conv4": 8, "Conv5": 16, "Up_conv5": 8,
                   "Up_conv4": 4, "Up_conv3": 2, "Up_conv2": 1}
ENCODER = ("Conv1", "Conv2", "Conv3", "Conv4", "Conv5")


def channel_dim(name: str, max_channel: int) -> int:
    return int(LAYER_DIMENSION[name] / 16 * max_channel)


def init_state_dict(input_dim: int, num_classes: int, max_channel: int, seed: int = 0) -> Dict[str, Tensor]:
    """Random (seeded) parameters with the reference's names/shapes; BN buffers at their defaults."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def conv(name, cout, cin, k=3):
        fan_in = cin * k * k
        sd[name] = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * (1.0 / fan_in) ** 0.5 * 1.7

    def bn(prefix, c):
        sd[prefix + ".weight"] = 0.5 + torch.rand(c, generator=g)
        sd[prefix + ".bias"] = (torch.rand(c, generator=g) - 0.5) * 0.4
        sd[prefix + ".running_mean"] = torch.zeros(c)
        sd[prefix + ".running_var"] = torch.ones(c)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def block(name, cin, cout):
        conv(f"_{name}.conv.0.weight", cout, cin)
        bn(f"_{name}.conv.1", cout)
        conv(f"_{name}.conv.3.weight", cout, cout)
        bn(f"_{name}.conv.4", cout)

    def up(name, cin, cout):
        conv(f"_{name}.up.1.weight", cout, cin)
        bn(f"_{name}.up.2", cout)

    ch = lambda n: channel_dim(n, max_channel)  # noqa: E731
    block("Conv1", input_dim, ch("Conv1"))
    block("Conv2", ch("Conv1"), ch("Conv2"))
    block("Conv3", ch("Conv2"), ch("Conv3"))
    block("Conv4", ch("Conv3"), ch("Conv4"))
    block("Conv5", ch("Conv4"), ch("Conv5"))
    up("Up5", ch("Conv5"), ch("Up_conv5"))
    block("Up_conv5", ch("Conv5"), ch("Up_conv5"))
    up("Up4", ch("Up_conv5"), ch("Up_conv4"))
    block("Up_conv4", ch("Up_conv5"), ch("Up_conv4"))
    up("Up3", ch("Up_conv4"), ch("Up_conv3"))
    block("Up_conv3", ch("Up_conv4"), ch("Up_conv3"))
    up("Up2", ch("Up_conv3"), ch("Up_conv2"))
    block("Up_conv2", ch("Up_conv3"), ch("Up_conv2"))
    conv("_Deconv_1x1.weight", num_classes, ch("Up_conv2"), k=1)
    sd["_Deconv_1x1.bias"] = (torch.rand(num_classes, generator=g) - 0.5) * 0.2
    return sd


def _take_forced(force: Optional[dict], prefix: str):
    """next recorded evaluation of the BN layer `prefix` (see `unet_forward(force=)`), or None"""
    if force is None:
        return None
    pos = force.setdefault("_pos", {})
    i = pos.get(prefix, 0)
    pos[prefix] = i + 1
    return force[prefix][i]


def _force_raw(y: Tensor, e: Optional[dict]) -> Tensor:
    """value of the device's raw conv output, gradient path of the exact one"""
    if e is None:
        return y
    return y + (e["y"].to(y.dtype) - y).detach()


def _bn_relu(sd, prefix: str, y: Tensor, training: bool, momentum: float, track: bool,
             round_dtype: Optional[torch.dtype], forced: Optional[dict] = None) -> Tensor:
    """nn.BatchNorm2d(momentum) + ReLU (unet.py:22-23): biased batch variance for the
    normalisation, unbiased for the running estimate, eps 1e-5.
    `forced` (a record of the device's own evaluation: raw output y, BN scale / shift, block output a)
    pins the ReLU routing to the device's decisions and the activation VALUES to the device's, so that
    the derivative computed here is the exact derivative of the function the device evaluated."""
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    use_batch = training
    out = F.batch_norm(y, rm if (not training or track) else None, rv if (not training or track) else None,
                       sd[prefix + ".weight"], sd[prefix + ".bias"], use_batch, momentum, 1e-5)
    if training and track:
        sd[prefix + ".num_batches_tracked"] += 1
    if forced is not None:
        # the device routes by fmaf(scale, y, shift) > 0 in f32; the f64 product of two f32 values is
        # exact and the f64 sum cannot change the sign, so this is the same predicate
        pre = (forced["scale"].double().view(1, -1, 1, 1) * forced["y"].double()
               + forced["shift"].double().view(1, -1, 1, 1))
        out = out * (pre > 0).to(out.dtype)
        a_dev = forced["a"] if forced.get("a") is not None else pre.clamp_min(0)
        return out + (a_dev.to(out.dtype) - out).detach()
    out = F.relu(out)
    if round_dtype is not None:
        out = out.to(round_dtype).to(y.dtype)
    return out


def _round(t: Tensor, round_dtype: Optional[torch.dtype]) -> Tensor:
    return t if round_dtype is None else t.to(round_dtype).to(t.dtype)


def _conv_block(sd, name: str, x: Tensor, training, momentum, track, rd, feats, force=None) -> Tensor:
    w0 = _round(sd[f"_{name}.conv.0.weight"], rd)
    e = _take_forced(force, f"_{name}.conv.1")
    y = _force_raw(_round(F.conv2d(x, w0, None, 1, 1), rd), e)
    a = _bn_relu(sd, f"_{name}.conv.1", y, training, momentum, track, rd, e)
    w1 = _round(sd[f"_{name}.conv.3.weight"], rd)
    e = _take_forced(force, f"_{name}.conv.4")
    y = _force_raw(_round(F.conv2d(a, w1, None, 1, 1), rd), e)
    a = _bn_relu(sd, f"_{name}.conv.4", y, training, momentum, track, rd, e)
    feats[name] = a
    return a


def _up_conv(sd, name: str, x: Tensor, training, momentum, track, rd, feats, force=None) -> Tensor:
    x = F.interpolate(x, scale_factor=2, mode="nearest")  # nn.Upsample(scale_factor=2) default
    w = _round(sd[f"_{name}.up.1.weight"], rd)
    e = _take_forced(force, f"_{name}.up.2")
    y = _force_raw(_round(F.conv2d(x, w, None, 1, 1), rd), e)
    a = _bn_relu(sd, f"_{name}.up.2", y, training, momentum, track, rd, e)
    feats[name] = a
    return a


def unet_forward(sd: Dict[str, Tensor], x: Tensor, *, training: bool = True, momentum: float = 0.1,
                 track_running_stats: bool = True, until: Optional[str] = None,
                 round_dtype: Optional[torch.dtype] = None, feats: Optional[dict] = None,
                 force: Optional[dict] = None) -> Tensor:
    """UNet.forward(x, until) of the reference.  `sd` BN buffers are updated in place in training
    mode.  round_dtype=torch.bfloat16 emulates the production storage precision (weights, raw
    conv outputs and activations rounded to bf16, arithmetic in f32).
    `force` = {BN prefix ("_Conv1.conv.1", "_Up5.up.2", ...): [record per evaluation, in call order]}
    with records {"y": raw conv output, "scale", "shift": the BN coefficients, "a": block output or None}
    taken from the device's own forward pass (cyhip.functions.RAW_TAP): values and ReLU / max-pool
    routing are then the device's, the derivative is exact -- the deterministic gradient-parity mode
    (a pre-activation that rounds to the other side of zero can no longer re-route a gradient)."""
    feats = {} if feats is None else feats
    t, m, k, rd = training, momentum, track_running_stats, round_dtype
    if force is not None:
        import functools
        cb = functools.partial(_conv_block, force=force)
        uc = functools.partial(_up_conv, force=force)
        return _unet_body(sd, x, until, cb, uc, t, m, k, rd, feats)
    return _unet_body(sd, x, until, _conv_block, _up_conv, t, m, k, rd, feats)


def _unet_body(sd, x, until, _conv_block, _up_conv, t, m, k, rd, feats):
    e1 = _conv_block(sd, "Conv1", x, t, m, k, rd, feats)
    if until == "Conv1":
        return e1
    e2 = _conv_block(sd, "Conv2", F.max_pool2d(e1, 2, 2), t, m, k, rd, feats)
    if until == "Conv2":
        return e2
    e3 = _conv_block(sd, "Conv3", F.max_pool2d(e2, 2, 2), t, m, k, rd, feats)
    if until == "Conv3":
        return e3
    e4 = _conv_block(sd, "Conv4", F.max_pool2d(e3, 2, 2), t, m, k, rd, feats)
    if until == "Conv4":
        return e4
    e5 = _conv_block(sd, "Conv5", F.max_pool2d(e4, 2, 2), t, m, k, rd, feats)
    if until == "Conv5":
        return e5
    d5 = _up_conv(sd, "Up5", e5, t, m, k, rd, feats)
    d5 = _conv_block(sd, "Up_conv5", torch.cat((e4, d5), dim=1), t, m, k, rd, feats)
    if until == "Up_conv5":
        return d5
    d4 = _up_conv(sd, "Up4", d5, t, m, k, rd, feats)
    d4 = _conv_block(sd, "Up_conv4", torch.cat((e3, d4), dim=1), t, m, k, rd, feats)
    if until == "Up_conv4":
        return d4
    d3 = _up_conv(sd, "Up3", d4, t, m, k, rd, feats)
    d3 = _conv_block(sd, "Up_conv3", torch.cat((e2, d3), dim=1), t, m, k, rd, feats)
    if until == "Up_conv3":
        return d3
    d2 = _up_conv(sd, "Up2", d3, t, m, k, rd, feats)
    d2 = _conv_block(sd, "Up_conv2", torch.cat((e1, d2), dim=1), t, m, k, rd, feats)
    if until == "Up_conv2":
        return d2
    logits = F.conv2d(d2, sd["_Deconv_1x1.weight"], sd["_Deconv_1x1.bias"])
    feats["Deconv_1x1"] = logits
    return logits


def clone_state_dict(sd: Dict[str, Tensor], requires_grad: bool = False, dtype=None) -> Dict[str, Tensor]:
    out = {}
    for k, v in sd.items():
        v = v.detach().clone()
        if v.is_floating_point():
            if dtype is not None:
                v = v.to(dtype)
            if requires_grad and "running" not in k:
                v.requires_grad_(True)
        out[k] = v
    return out
