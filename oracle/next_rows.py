"""ORACLE (test infrastructure, not product code): CPU restatements of the SURVEY.md section-8
rows built after the U-Net/InfoNCE core -- dense InfoNCE point sampling (a8), cluster heads and
discrete mutual-information losses (a17), the GroupNorm+SiLU block (a18), bilinear resize (a19)
and the warm-up/cosine learning-rate law.  Every function cites the reference lines it follows;
tests/test_oracle_golden.py pins them against tests/golden/next_rows.npz, produced by importing
the reference's own modules (tests/golden/gen_goldens.py --only next).
"""
from __future__ import annotations

import math
from typing import Optional, Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


# ---------------------------------------------------------------- a8: dense InfoNCE sampling
def region_points(n: int, h: int, w: int, seed: int, point_nums: int = 5) -> List[List[Tuple[int, int]]]:
    """semi_seg/hooks/infonce.py:31-46: after seeding python/numpy/torch with `seed`, for every
    image in order draw `point_nums` distinct rows then `point_nums` distinct columns with
    np.random.choice(range(.), replace=False) and zip them.  (`fix_all_seed_for_transforms`
    = random.seed, np.random.seed, torch.manual_seed; only numpy's stream is consumed.)"""
    rs = np.random.RandomState(seed)
    pts = []
    for _ in range(n):
        hs = rs.choice(range(h), point_nums, replace=False)
        ws = rs.choice(range(w), point_nums, replace=False)
        pts.append([(int(a), int(b)) for a, b in zip(hs, ws)])
    return pts


def region_extractor(feat: Tensor, seed: int, point_nums: int = 5) -> Tensor:
    """[n,D,h,w] -> [point_nums*n, D]: image-major, point order as drawn (infonce.py:36-46)"""
    n, _, h, w = feat.shape
    pts = region_points(n, h, w, seed, point_nums)
    return torch.cat([torch.stack([feat[i, :, a, b] for a, b in pts[i]], dim=0) for i in range(n)], dim=0)


# ---------------------------------------------------------------- a17: cluster heads + MI
def cluster_head(sds: Sequence[Dict[str, Tensor]], feat: Tensor, T: float = 1.0) -> List[Tensor]:
    """projectors/heads.py:44-62,125-148 (`head_type="linear"`, normalize=False): per sub-head
    global average pool -> Linear(C, k) -> softmax(logits / T)"""
    pooled = feat.mean(dim=(2, 3))
    return [F.softmax(F.linear(pooled, sd["2.weight"], sd["2.bias"]) / T, dim=1) for sd in sds]


def dense_cluster_head(sds: Sequence[Dict[str, Tensor]], feat: Tensor, T: float = 1.0) -> List[Tensor]:
    """projectors/heads.py:65-78,151-173 (`head_type="linear"`, normalize=False): per sub-head
    Conv1x1(C, k) -> softmax over channels of logits / T"""
    return [F.softmax(F.conv2d(feat, sd["0.weight"], sd["0.bias"]) / T, dim=1) for sd in sds]


def init_cluster_sds(input_dim: int, k: int, subheads: int, dense: bool, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(subheads):
        w = (torch.rand(k, input_dim, generator=g) * 2 - 1) / math.sqrt(input_dim)
        b = (torch.rand(k, generator=g) * 2 - 1) / math.sqrt(input_dim)
        out.append({"0.weight": w.view(k, input_dim, 1, 1), "0.bias": b} if dense else {"2.weight": w, "2.bias": b})
    return out


def joint_vectors(x: Tensor, y: Tensor, symmetric: bool = True) -> Tensor:
    """losses/discreteMI.py:201-222: k x k joint of two [n,k] simplex batches, symmetrised,
    normalised to sum 1"""
    p = (x.unsqueeze(2) * y.unsqueeze(1)).sum(0)
    if symmetric:
        p = (p + p.t()) / 2.0
    return p / p.sum()


def iid_loss(x: Tensor, y: Tensor, lamb: float = 1.0) -> Tuple[Tensor, Tensor, Tensor]:
    """losses/discreteMI.py:90-124 `IIDLoss.forward`: (-MI with lambda, -MI, joint)"""
    k = x.shape[1]
    p = joint_vectors(x, y)
    pi = p.sum(1).view(k, 1).expand(k, k)
    pj = p.sum(0).view(1, k).expand(k, k)
    lp, li, lj = torch.log(p + 1e-10), torch.log(pi + 1e-10), torch.log(pj + 1e-10)
    return (-p * (lp - lamb * lj - lamb * li)).sum(), (-p * (lp - lj - li)).sum(), p


def joint_maps(x: Tensor, y: Tensor, padding: int = 0, symmetric: bool = False) -> Tensor:
    """[n,k,H,W] x2 -> [T,T,k,k], T = 2*padding+1.
    padding == 0 (losses/discreteMI.py:246-261): (X/sqrt(N)) (Y/sqrt(N))^T over all N=n*H*W
    pixels, optionally symmetrised, NOT renormalised.
    padding  > 0 (losses/discreteMI.py:225-243): displaced co-occurrence -- conv2d with the
    first map as input [k,n,H,W] and the second as weight [k,n,H,W]; subtract the global min,
    add 1e-8, normalise each displacement to sum 1, optionally symmetrise, normalise globally."""
    k = x.shape[1]
    if padding == 0:
        xf = x.transpose(0, 1).reshape(k, -1)
        yf = y.transpose(0, 1).reshape(k, -1)
        n = xf.shape[1]
        p = (xf / math.sqrt(n)) @ (yf.t() / math.sqrt(n))
        if symmetric:
            p = (p + p.t()) / 2.0
        return p.view(1, 1, k, k)
    p = F.conv2d(x.transpose(0, 1).contiguous(), y.transpose(0, 1).contiguous(), padding=(padding, padding))
    p = p - p.min().detach() + 1e-8
    p = p.permute(2, 3, 0, 1)
    p = p / p.sum(dim=[2, 3], keepdim=True)
    if symmetric:
        p = (p + p.permute(0, 1, 3, 2)) / 2.0
    return p / p.sum()


def iid_segmentation_loss(x: Tensor, y: Tensor, lamda: float = 1.0, padding: int = 0, eps: float = 1e-5,
                          symmetric: bool = False) -> Tensor:
    """losses/discreteMI.py:127-170 `IIDSegmentationLoss.forward` (mask=None)"""
    T = 2 * padding + 1
    p = joint_maps(x, y, padding, symmetric)
    pi = p.sum(dim=2, keepdim=True)
    pj = p.sum(dim=3, keepdim=True)
    loss = -p * (torch.log(p + eps) - lamda * torch.log(pi + eps) - lamda * torch.log(pj + eps))
    return loss.sum() / (T * T)


# ---------------------------------------------------------------- a18: GroupNorm + SiLU block
def gn_silu_block(x: Tensor, weight: Tensor, bias: Tensor, gamma: Tensor, beta: Tensor, groups: int = 8,
                  eps: float = 1e-5) -> Tensor:
    """arch/unet2.py:208-224 `Block.forward` without scale_shift:
    Conv2d(3x3, padding 1, bias) -> GroupNorm(groups, C) -> SiLU"""
    y = F.conv2d(x, weight, bias, padding=1)
    return F.silu(F.group_norm(y, groups, gamma, beta, eps))


def init_gn_block(cin: int, cout: int, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    bound = 1.0 / math.sqrt(cin * 9)
    return {"proj.weight": (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) * bound,
            "proj.bias": (torch.rand(cout, generator=g) * 2 - 1) * bound,
            "norm.weight": 0.5 + torch.rand(cout, generator=g),
            "norm.bias": torch.rand(cout, generator=g) - 0.5}


# ---------------------------------------------------------------- a19: bilinear resize
def bilinear_resize(x: Tensor, size: Sequence[int]) -> Tensor:
    """semi_seg/hooks/cc.py:132 / ccblock.py:300: F.interpolate(mode="bilinear"), default
    align_corners=False, no antialias: src = (dst + 0.5) * in/out - 0.5 clamped at 0, the upper
    neighbour clamped to the last row/column"""
    n, c, H, W = x.shape
    h, w = size

    def taps(out_len, in_len):
        s = (torch.arange(out_len, dtype=torch.float64) + 0.5) * (in_len / out_len) - 0.5
        s = s.clamp(min=0)
        i0 = s.floor().long().clamp(max=in_len - 1)
        i1 = (i0 + 1).clamp(max=in_len - 1)
        f = (s - i0.double()).to(x.dtype)
        return i0, i1, f

    y0, y1, fy = taps(h, H)
    x0, x1, fx = taps(w, W)
    top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
    bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
    return top * (1 - fy).view(1, 1, h, 1) + bot * fy.view(1, 1, h, 1)


# ---------------------------------------------------------------- lr schedule
def warmup_cosine_lrs(base_lr: float, multiplier: float, warmup: int, max_epoch: int, eta_min: float = 1e-7,
                      epochs: int = None) -> List[float]:
    """lr seen by epoch e = 0.. (one scheduler.step() per epoch), contrastyou/optim/scheduler.py:56-72
    + trainer/base.py:77-89: linear from base_lr to base_lr*multiplier over `warmup` epochs, then
    CosineAnnealingLR(T_max=max_epoch-warmup, eta_min) whose base lr is base_lr*multiplier and whose
    own epoch counter starts at 0 on the first post-warm-up call and advances by one from then on
    (torch's chained/recursive cosine form, evaluated from the current lr)."""
    epochs = max_epoch if epochs is None else epochs
    t_max = max_epoch - warmup
    peak = base_lr * multiplier
    lrs, lr, cos_epoch, finished = [], base_lr, 0, False
    for e in range(epochs):
        lrs.append(lr)
        nxt = e + 1
        if nxt <= warmup:
            lr = base_lr * ((multiplier - 1.0) * nxt / warmup + 1.0)
        elif not finished:
            finished = True  # hand-over: get_lr() of the cosine scheduler at ITS epoch 0
            lr = _cosine_chain(lr, cos_epoch, t_max, eta_min, peak)
        else:
            cos_epoch += 1
            lr = _cosine_chain(lr, cos_epoch, t_max, eta_min, peak)
    return lrs


def _cosine_chain(lr: float, t: int, t_max: int, eta_min: float, base: float) -> float:
    """torch.optim.lr_scheduler.CosineAnnealingLR.get_lr, recursive form as shipped in torch 2.10: at
    the hand-over it is evaluated with t = 0 (no `last_epoch == 0` special case outside __init__),
    which lifts the peak by the factor 2 / (1 + cos(pi / t_max))"""
    if (t - 1 - t_max) % (2 * t_max) == 0:
        return lr + (base - eta_min) * (1 - math.cos(math.pi / t_max)) / 2
    return (1 + math.cos(math.pi * t / t_max)) / (1 + math.cos(math.pi * (t - 1) / t_max)) * (lr - eta_min) + eta_min


# ---------------------------------------------------------------- round 2: f2 / f4 leftovers
def cluster_head_general(sds: Sequence[Dict[str, Tensor]], feat: Tensor, *, dense: bool, head_type: str,
                         normalize: bool, T: float = 1.0) -> List[Tensor]:
    """projectors/heads.py:44-78 `init_sub_header` / `init_dense_sub_header` in all four variants:
    [pool ->] Linear/Conv1x1 [-> LeakyReLU(0.01) -> Linear/Conv1x1] [-> F.normalize(dim=1)] -> softmax(./T).
    `sds[i]` is sub-head i's state dict with the Sequential's own indices as keys."""
    outs = []
    for sd in sds:
        if dense:
            i0, i1 = ("0", "2")
            x = F.conv2d(feat, sd[f"{i0}.weight"], sd[f"{i0}.bias"])
            if head_type == "mlp":
                x = F.conv2d(F.leaky_relu(x, 0.01), sd[f"{i1}.weight"], sd[f"{i1}.bias"])
        else:
            i0, i1 = ("2", "4")
            x = F.linear(feat.mean(dim=(2, 3)), sd[f"{i0}.weight"], sd[f"{i0}.bias"])
            if head_type == "mlp":
                x = F.linear(F.leaky_relu(x, 0.01), sd[f"{i1}.weight"], sd[f"{i1}.bias"])
        if normalize:
            x = F.normalize(x, p=2, dim=1)
        outs.append(F.softmax(x / T, dim=1))
    return outs


def redundancy_criterion(x_out: Tensor, x_tf_out: Tensor, *, alpha: float, lamda: float = 1.0, eps: float = 1e-5,
                         symmetric: bool = True) -> Tensor:
    """losses/redundancy_reduction.py:21-33 on the joint of losses/discreteMI.py:246-261"""
    k = x_out.shape[1]
    a = x_out.swapaxes(0, 1).reshape(k, -1)
    b = x_tf_out.swapaxes(0, 1).reshape(k, -1)
    n = a.shape[1]
    p = (a / math.sqrt(n)) @ (b.t() / math.sqrt(n))
    if symmetric:
        p = (p + p.t()) / 2.0
    target = torch.eye(k, dtype=p.dtype) / k * alpha + p * (1 - alpha)
    p_i = p.sum(dim=1).view(k, 1).expand(k, k)
    p_j = p.sum(dim=0).view(1, k).expand(k, k)
    constrained = (-p * (-lamda * torch.log(p_j + eps) - lamda * torch.log(p_i + eps))).sum()
    return -(target * (p + eps).log()).sum() + constrained


def self_paced_supcon(z1: Tensor, z2: Tensor, target: Optional[Sequence[int]] = None, *, gamma: float,
                      weight_update: str = "hard", correct_grad: bool = False, t: float = 0.07,
                      mask: Optional[Tensor] = None):
    """losses/contrastive.py:103-212: SupConLoss1 with the positives weighted by a no-grad self-paced mask;
    returns (loss, downgrade ratio).  With an explicit `mask` the positives are mask == 1 and the negatives
    mask == 0 (:117-121): any other value (e.g. -1) drops the pair from both."""
    n = z1.shape[0]
    if mask is not None:
        pos, neg = (mask == 1).to(z1.dtype), (mask == 0).to(z1.dtype)
    else:
        lab = torch.as_tensor(list(target))
        pos = torch.eq(lab[:, None], lab[None, :]).to(z1.dtype)
        neg = 1 - pos
    off = 1 - torch.eye(2 * n, dtype=z1.dtype)
    pos_mask, neg_mask = pos.repeat(2, 2) * off, neg.repeat(2, 2) * off
    P = torch.cat([z1, z2], 0)
    sim = P @ P.t() / t
    logits = sim - sim.max().detach()
    e = torch.exp(logits)
    denom = (e * pos_mask).sum(1, keepdim=True) + (e * neg_mask).sum(1, keepdim=True)
    llh = logits - torch.log(denom + 1e-16)
    with torch.no_grad():
        l = -llh
        w = (l <= gamma).to(z1.dtype) if weight_update == "hard" else torch.clamp(1 - l / gamma, min=0)
        sp = torch.max(w, 1 - pos_mask)
    ratio = torch.masked_select(sp, pos_mask.bool()).mean().item()
    loss = -(((llh * sp) * pos_mask).sum(1) / pos_mask.sum(1)).mean()
    if correct_grad and ratio > 0:
        loss = loss / ratio
    return loss, ratio
