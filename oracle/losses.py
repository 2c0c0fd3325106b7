"""ORACLE (test infrastructure, not product code): CPU restatements of the reference's loss /
projector / metric leaves on the SemiSupervisedEpocher hot path.  Each function cites the
reference lines it follows; tests/test_oracle_golden.py pins them against vectors produced by
importing /root/reference (tests/golden/gen_goldens.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor


# ---------------------------------------------------------------- supervised loss
def class2one_hot(seg: Tensor, C: int) -> Tensor:
    """contrastyou/utils/general.py:114-120: F.one_hot(seg, C) moved to dim 1"""
    return F.one_hot(seg.long(), C).movedim(-1, 1)


def kl_div(prob: Tensor, target: Tensor, eps: float = 1e-16) -> Tensor:
    """contrastyou/losses/kl.py:112-125 with reduction='mean', no class weights:
    mean over (n,h,w) of sum_c -t*log((p+eps)/(t+eps))"""
    kl = -target * torch.log((prob + eps) / (target + eps))
    return kl.sum(1).mean()


def sup_loss(logits: Tensor, target: Tensor, eps: float = 1e-16) -> Tensor:
    """semi_seg/epochers/epocher.py:317-318: KL_div(softmax(logits,1), one_hot(target))"""
    C = logits.shape[1]
    return kl_div(logits.softmax(1), class2one_hot(target, C).to(logits.dtype), eps)


def softmax_mse(a: Tensor, b: Tensor) -> Tensor:
    """semi_seg/hooks/consistency.py:36 / mt.py:186: nn.MSELoss()(a.softmax(1), b.softmax(1))"""
    return F.mse_loss(a.softmax(1), b.softmax(1))


# ---------------------------------------------------------------- projection heads
def projection_head(sd: Dict[str, Tensor], feat: Tensor, normalize: bool = True, pool: str = "adaptive_avg") -> Tensor:
    """contrastyou/projectors/heads.py:14-22 (head_type='mlp'):
    AdaptiveAvgPool2d(1) -> Flatten -> Linear -> LeakyReLU(0.01) -> Linear -> F.normalize(dim=1)
    (pool="adaptive_max": nn.AdaptiveMaxPool2d(1) instead, projectors/nn.py:16-23)
    sd keys: '_header.2.weight/bias', '_header.4.weight/bias' (nn.Sequential indices)."""
    x = feat.mean(dim=(2, 3)) if pool == "adaptive_avg" else feat.amax(dim=(2, 3))
    x = F.linear(x, sd["_header.2.weight"], sd["_header.2.bias"])
    x = F.leaky_relu(x, 0.01)
    x = F.linear(x, sd["_header.4.weight"], sd["_header.4.bias"])
    return F.normalize(x, p=2, dim=1) if normalize else x


def dense_projection_head(sd: Dict[str, Tensor], feat: Tensor, spatial_size: Sequence[int],
                          normalize: bool = True, pool: str = "adaptive_avg") -> Tensor:
    """contrastyou/projectors/heads.py:33-38,112-119: conv1x1 -> LeakyReLU -> conv1x1 ->
    AdaptiveAvgPool2d(spatial_size) (or AdaptiveMaxPool2d, pool="adaptive_max") -> F.normalize over channels."""
    x = F.conv2d(feat, sd["_projector.0.weight"], sd["_projector.0.bias"])
    x = F.leaky_relu(x, 0.01)
    x = F.conv2d(x, sd["_projector.2.weight"], sd["_projector.2.bias"])
    x = (F.adaptive_avg_pool2d if pool == "adaptive_avg" else F.adaptive_max_pool2d)(x, tuple(spatial_size))
    return F.normalize(x, p=2, dim=1) if normalize else x


# ---------------------------------------------------------------- InfoNCE / SupCon
def supcon_masks(n: int, target: Optional[Sequence[int]] = None, mask: Optional[Tensor] = None):
    """contrastyou/losses/contrastive.py:31-50,62-71: positive / negative masks, tiled 2x2,
    diagonal removed."""
    if mask is not None:
        pos = (mask == 1).float()
    elif target is not None:
        t = torch.as_tensor(list(target), dtype=torch.float32)
        pos = torch.eq(t[:, None], t[None, :]).float()
    else:
        pos = torch.eye(n)
    neg = 1 - pos
    off = 1 - torch.eye(2 * n)
    return pos.repeat(2, 2) * off, neg.repeat(2, 2) * off


def supcon_loss_exclude_pos(z1: Tensor, z2: Tensor, target: Optional[Sequence[int]] = None,
                            mask: Optional[Tensor] = None, t: float = 0.07) -> Tensor:
    """SupConLoss1(exclude_other_pos=True) (contrastive.py:84-91): per positive pair the denominator holds that pair and
    the negatives, the negatives' sum divided by (neg / (pos + neg) + 1e-4):
    loss = -mean_i[ sum_j pos_ij (S_ij - log(E_ij + negsum_i / (negratio_i + 1e-4) + 1e-16)) / sum_j pos_ij ]"""
    n = z1.shape[0]
    pos, neg = supcon_masks(n, target, mask)
    pos, neg = pos.to(z1), neg.to(z1)
    P = torch.cat([z1, z2], dim=0)
    S = P @ P.t() / t
    S = S - S.max().detach()
    E = torch.exp(S)
    pos_count, neg_count = pos.sum(1), neg.sum(1)
    neg_sum = (E * neg).sum(1, keepdim=True)
    ratio = (neg_count / (pos_count + neg_count))[:, None]
    log_ratio = S - torch.log(E + neg_sum / (ratio + 1e-4) + 1e-16)
    return -((log_ratio * pos).sum(1) / pos_count).mean()


def supcon_loss(z1: Tensor, z2: Tensor, target: Optional[Sequence[int]] = None,
                mask: Optional[Tensor] = None, t: float = 0.07, return_all: bool = False):
    """SupConLoss1._forward (contrastive.py:52-100) with exp_sim_temperature (:14-20):
    S = P P^T / t;  S -= max(S).detach();  E = exp(S)
    loss = -mean_i[ sum_j pos_ij (S_ij - log(sum_k (pos+neg)_ik E_ik + 1e-16)) / sum_j pos_ij ]"""
    n = z1.shape[0]
    pos, neg = supcon_masks(n, target, mask)
    pos, neg = pos.to(z1), neg.to(z1)
    P = torch.cat([z1, z2], dim=0)
    S = P @ P.t() / t
    S = S - S.max().detach()
    E = torch.exp(S)
    pos_count = pos.sum(1)
    denom = (E * pos).sum(1, keepdim=True) + (E * neg).sum(1, keepdim=True)
    log_ratio = S - torch.log(denom + 1e-16)
    loss = -((log_ratio * pos).sum(1) / pos_count).mean()
    if return_all:
        return loss, S, E, pos, neg
    return loss


# ---------------------------------------------------------------- label generators
def encode_labels(values: Sequence[str]) -> List[int]:
    """sklearn LabelEncoder().fit(v).transform(v): rank among the sorted unique values
    (semi_seg/epochers/helper.py:54-62)"""
    uniq = sorted(set(values))
    lut = {v: i for i, v in enumerate(uniq)}
    return [lut[v] for v in values]


def get_label(contrast_on: str, data_name: str, partition_group: Sequence[str],
              label_group: Sequence[str]) -> List[int]:
    """semi_seg/hooks/utils.py:74-102 + helper.py:54-71"""
    if contrast_on == "partition":
        return encode_labels(list(partition_group))
    if contrast_on == "patient":
        if "acdc" in data_name or data_name in ("prostate", "prostate_md"):
            return encode_labels([p.split("_")[0] for p in label_group])
        return encode_labels(list(label_group))
    if contrast_on == "cycle":
        return [0 if p.split("_")[1] == "00" else 1 for p in label_group]
    if contrast_on == "self":
        return list(range(len(partition_group)))
    raise NotImplementedError(contrast_on)


# ---------------------------------------------------------------- Dice
def dice_summary(preds: List[Tensor], targets: List[Tensor], groups: List[List[str]], C: int,
                 report_axis: Sequence[int]) -> Dict[str, float]:
    """contrastyou/meters/general_dice_meter.py:37-91: per-group accumulated intersection /
    union of one-hot pred and target, dice = (2I+1e-16)/(U+1e-16), mean over groups, then
    DSC_mean over the reported classes."""
    inter: Dict[str, Tensor] = {}
    union: Dict[str, Tensor] = {}
    for pred, tgt, grp in zip(preds, targets, groups):
        po = class2one_hot(pred, C).long()
        to = class2one_hot(tgt, C).long()
        i_ = (po * to).sum(dim=(2, 3))
        u_ = (po + to).sum(dim=(2, 3))
        for b, g in enumerate(grp):
            inter[g] = inter.get(g, 0) + i_[b]
            union[g] = union.get(g, 0) + u_[b]
    I = torch.stack(list(inter.values()), 0).float()
    U = torch.stack(list(union.values()), 0).float()
    dices = (2 * I + 1e-16) / (U + 1e-16)
    means = dices.mean(0)
    out = {f"DSC{i}": float(means[i]) for i in report_axis}
    out["DSC_mean"] = sum(out.values()) / len(out)
    return out


# ---------------------------------------------------------------- affine augmentation
def affine_source_index(theta: Tensor, H: int, W: int):
    """(flat source index [N, H*W], valid mask) of the nearest-neighbour affine resampling: output pixel ->
    normalised coordinate -> theta -> source pixel, affine_grid / grid_sample conventions
    (align_corners=False, round-half-to-even, zeros outside).  The coordinate arithmetic is part of the
    definition (a nearest-neighbour pick is discontinuous): f64, one IEEE operation at a time in this order --
    the device kernel (csrc/cy_misc.hip affine_src) performs the same sequence bit for bit."""
    t = theta.detach().to(torch.float64)  # the f32 values of theta, exactly
    N = t.shape[0]
    xo = ((2 * torch.arange(W, dtype=torch.float64) + 1) / float(W) - 1.0).view(1, 1, W)
    yo = ((2 * torch.arange(H, dtype=torch.float64) + 1) / float(H) - 1.0).view(1, H, 1)
    c = lambda i, j: t[:, i, j].view(N, 1, 1)  # noqa: E731
    xi = (c(0, 0) * xo + c(0, 1) * yo) + c(0, 2)
    yi = (c(1, 0) * xo + c(1, 1) * yo) + c(1, 2)
    px = ((xi + 1.0) * float(W) - 1.0) * 0.5
    py = ((yi + 1.0) * float(H) - 1.0) * 0.5
    rx, ry = torch.round(px), torch.round(py)
    valid = (rx >= 0) & (rx <= W - 1) & (ry >= 0) & (ry <= H - 1)
    src = (ry.clamp(0, H - 1) * W + rx.clamp(0, W - 1)).long()
    return src.view(N, H * W), valid.view(N, H * W)


def affine_nearest(x: Tensor, theta: Tensor, gamma: Optional[Tensor] = None) -> Tensor:
    """The build's own definition of the in-step augmentation (the reference delegates to the
    un-vendored `rising`, semi_seg/augment.py:297-311 -- parity unpinned): optional gamma x**g, then
    nearest-neighbour resampling under theta with zeros padding -- F.grid_sample(F.affine_grid(theta),
    mode="nearest", align_corners=False) with the source coordinates computed in f64
    (`affine_source_index`), so that the pick is reproducible bit for bit on any device."""
    if gamma is not None:
        x = x ** gamma.view(-1, 1, 1, 1).to(x.dtype)
    N, C, H, W = x.shape
    src, valid = affine_source_index(theta, H, W)
    out = torch.gather(x.reshape(N, C, H * W), 2, src.view(N, 1, H * W).expand(N, C, H * W))
    return (out * valid.view(N, 1, H * W).to(x.dtype)).view(N, C, H, W)


def make_theta(scale: float, rot_deg: float, tx: float, ty: float, flip_h: bool, flip_w: bool) -> Tensor:
    """2x3 output->input matrix for scale / rotation / translation (+ mirror) in normalised coords,
    parameter ranges of semi_seg/epochers/epocher.py:227-237."""
    a = math.radians(rot_deg)
    c, s = math.cos(a) / scale, math.sin(a) / scale
    m = torch.tensor([[c, -s, tx], [s, c, ty]], dtype=torch.float32)
    if flip_w:
        m[:, 0] = -m[:, 0]
    if flip_h:
        m[:, 1] = -m[:, 1]
    return m


# ---------------------------------------------------------------- EMA / RAdam
def ema_update(teacher: Tensor, student: Tensor, alpha: float, weight_decay: float) -> Tensor:
    """semi_seg/hooks/mt.py:60-82: t = alpha*t + (1-alpha)*s ; t *= (1 - wd)"""
    return (alpha * teacher + (1 - alpha) * student) * (1 - weight_decay)


# ---------------------------------------------------------------- seeded parameter sets
def init_projector_sd(input_dim: int, hidden_dim: int, output_dim: int, seed: int = 0) -> Dict[str, Tensor]:
    """seeded weights with the reference's names for ProjectionHead (heads.py:81-96)"""
    g = torch.Generator().manual_seed(seed)

    def u(*shape, fan):
        return (torch.rand(*shape, generator=g) * 2 - 1) / math.sqrt(fan)

    return {"_header.2.weight": u(hidden_dim, input_dim, fan=input_dim),
            "_header.2.bias": u(hidden_dim, fan=input_dim),
            "_header.4.weight": u(output_dim, hidden_dim, fan=hidden_dim),
            "_header.4.bias": u(output_dim, fan=hidden_dim)}


def init_dense_projector_sd(input_dim: int, hidden_dim: int, output_dim: int, seed: int = 0) -> Dict[str, Tensor]:
    """seeded weights with the reference's names for DenseProjectionHead (heads.py:99-123)"""
    g = torch.Generator().manual_seed(seed)

    def u(*shape, fan):
        return (torch.rand(*shape, generator=g) * 2 - 1) / math.sqrt(fan)

    return {"_projector.0.weight": u(hidden_dim, input_dim, 1, 1, fan=input_dim),
            "_projector.0.bias": u(hidden_dim, fan=input_dim),
            "_projector.2.weight": u(output_dim, hidden_dim, 1, 1, fan=hidden_dim),
            "_projector.2.bias": u(output_dim, fan=hidden_dim)}
