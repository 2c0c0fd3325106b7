"""InfoNCE / supervised-contrastive loss with the interface of
contrastyou/losses/contrastive.py:23-100 (`SupConLoss1`), computed by the HIP SupCon kernels:
f32-MFMA similarity GEMM, row statistics with the reference's global-max shift, and the
(G + G^T) P / t backward -- see csrc/cy_contrast.hip.

    criterion(z1, z2, target=list|Tensor|None, mask=None) -> scalar loss
    criterion.sim_exp / .sim_logits / .pos_mask / .neg_mask   (materialised on first access)

The reference asserts unit-norm inputs (AssertionError) and raises RuntimeError on a NaN loss,
both through host syncs.  Here the same conditions are evaluated on the device and checked by
`validate()`; with `defer_checks=False` (default) validate() runs inside forward, exactly like the
reference; the training hooks switch to deferred mode and validate once per epoch.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import Tensor, nn

from cyhip import ops
from cyhip.functions import SgemmFn, SupConFn


def is_normalized(feature: Tensor, dim=1) -> bool:
    norms = feature.norm(dim=dim)
    return torch.allclose(norms, torch.ones_like(norms))


_label_cache = {}  # label tuple -> int32 device tensor (a few distinct partitions per dataset)


def _encode_target(target, n: int, device) -> Tensor:
    """labels as int32 codes (equality is all that matters: contrastive.py:41)"""
    if isinstance(target, Tensor):
        t = target.detach().reshape(-1)
        if t.is_floating_point():
            _, t = torch.unique(t, return_inverse=True)
        t = t.to(device=device, dtype=torch.int32)
    else:
        vals = tuple(target)
        assert len(vals) == n, (len(vals), n)
        key = (vals, str(device))
        t = _label_cache.get(key)
        if t is None:
            lut = {}
            host = torch.tensor([lut.setdefault(v, len(lut)) for v in vals], dtype=torch.int32)
            t = ops.pinned.upload(host, device)
            if len(_label_cache) > 512:
                _label_cache.clear()
            _label_cache[key] = t
        return t
    assert t.numel() == n, (t.numel(), n)
    return t.contiguous()


def _rejoin(a: Tensor, b: Tensor) -> Tensor:
    """cat([a, b]) -- without the copy (and, in the backward, without the zero-fill + two slice copies of the
    chunk's gradient) when a and b are the two halves `torch.chunk` made of one contiguous tensor, which is how the
    InfoNCE hook hands the projector's output over (semi_seg/hooks/infonce.py:226-230 in the reference)"""
    base = a._base
    if (base is not None and base is b._base and base.dim() == a.dim() and base.is_contiguous()
            and a.is_contiguous() and b.is_contiguous() and base.shape[0] == a.shape[0] + b.shape[0]
            and base.shape[1:] == a.shape[1:] == b.shape[1:] and a.data_ptr() == base.data_ptr()
            and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()):
        return base
    return torch.cat([a, b], dim=0)


class SupConLoss1(nn.Module):
    def __init__(self, temperature=0.07, exclude_other_pos=False):
        super().__init__()
        self._t = temperature
        self._exclude_pos = exclude_other_pos
        self.defer_checks = False
        self._pending = []
        self._last = None

    def forward(self, proj_feat1: Tensor, proj_feat2: Tensor, target=None, mask: Optional[Tensor] = None, **kwargs):
        assert proj_feat1.shape == proj_feat2.shape, (proj_feat1.shape, proj_feat2.shape)
        ops.require_gpu(proj_feat1, proj_feat2)
        n = proj_feat1.size(0)
        dev = proj_feat1.device
        labels = pos = None
        if mask is not None:
            assert mask.shape == torch.Size([n, n])
            pos = (mask == 1).to(device=dev, dtype=torch.uint8).contiguous()
        elif target is not None:
            labels = _encode_target(target, n, dev)
        else:
            labels = torch.arange(n, dtype=torch.int32, device=dev)  # SimCLR: only the other view
        P = _rejoin(proj_feat1, proj_feat2)
        loss, diag, stats = SupConFn.apply(P, labels, pos, float(self._t), self._exclude_pos)
        self._last = (P.detach(), stats, labels, pos)
        # device-side evidence for the reference's two checks: one aminmax launch now, the arithmetic in validate()
        mn, mx = torch.aminmax(diag)
        self._pending.append((mn, mx, loss.detach()))
        if not self.defer_checks:
            self.validate()
        return loss

    def validate(self):
        """raise like the reference: AssertionError (inputs not unit norm), RuntimeError (NaN loss)"""
        pending, self._pending = self._pending, []
        if not pending:
            return
        ends = torch.stack([p[0] for p in pending] + [p[1] for p in pending])  # extreme S_ii = |P_i|^2 / t
        errs = (ends * self._t - 1).abs().max()
        losses = torch.stack([p[2] for p in pending])
        bad_norm, has_nan = (errs > 1e-4).item(), torch.isnan(losses).any().item()
        assert not bad_norm, "features need to be normalized first"
        if has_nan:
            raise RuntimeError(losses)

    def _matrices(self):
        if self._last is None:
            raise AttributeError("call the criterion first")
        P, stats, labels, pos = self._last  # (the matrices are inspection aids: the similarity is formed on demand)
        S = ops.sgemm(P.float().contiguous(), P.float().contiguous(), 1.0 / float(self._t), b_trans=True)
        return ops.supcon_matrices(S, stats, labels, pos)

    @property
    def sim_logits(self) -> Tensor:
        return self._matrices()[0]

    @property
    def sim_exp(self) -> Tensor:
        return self._matrices()[1]

    @property
    def pos_mask(self) -> Tensor:
        return self._matrices()[2]

    @property
    def neg_mask(self) -> Tensor:
        return self._matrices()[3]


class SelfPacedSupConLoss(nn.Module):
    """`SelfPacedSupConLoss` (contrastyou/losses/contrastive.py:103-212): SupConLoss1 whose positive pairs are
    weighted by a self-paced mask computed (without gradient) from their own log-likelihood,
    hard: [l_ij <= gamma], soft: max(1 - l_ij / gamma, 0); gamma -> infinity recovers SupConLoss1 (the identity
    the reference checks in its __main__, :241-248).  The similarity GEMM P P^T / t runs on the f32 MFMA kernel
    (cyhip.functions.SgemmFn, differentiable); the 2n x 2n elementwise part is plain autograd -- this loss is
    used at a few dozen rows by SelfPacedINFONCEHook, never at C5 sizes."""

    def __init__(self, temperature=0.07, weight_update="hard", correct_grad=False, **kwargs):
        super().__init__()
        assert weight_update in ("hard", "soft"), weight_update
        self._t, self._weight_update, self._correct_grad = temperature, weight_update, correct_grad
        self._gamma = 1e6

    def __repr__(self):
        return f"{self.__class__.__name__} with T: {self._t}, method: {self._weight_update} gamma: {self._gamma}"

    def set_gamma(self, gamma):
        self._gamma = float(gamma)

    @property
    def age_param(self):
        return self._gamma

    def forward(self, proj_feat1: Tensor, proj_feat2: Tensor, target=None, mask: Optional[Tensor] = None, **kwargs):
        assert proj_feat1.shape == proj_feat2.shape, (proj_feat1.shape, proj_feat2.shape)
        ops.require_gpu(proj_feat1, proj_feat2)
        n, dev = proj_feat1.shape[0], proj_feat1.device
        if mask is not None:
            assert mask.shape == torch.Size([n, n])
            pos = (mask == 1).float().to(dev)
            neg = (mask == 0).float().to(dev)  # (contrastive.py:120-121: other values, e.g. -1, are neither)
        elif target is not None:
            t = _encode_target(target, n, dev)
            pos = torch.eq(t[:, None], t[None, :]).float()
            neg = 1 - pos
        else:
            pos = torch.eye(n, dtype=torch.float, device=dev)
            neg = 1 - pos
        R = 2 * n
        off_diag = 1 - torch.eye(R, dtype=torch.float, device=dev)
        pos_mask, neg_mask = pos.repeat(2, 2) * off_diag, neg.repeat(2, 2) * off_diag
        P = torch.cat([proj_feat1, proj_feat2], dim=0)
        sim = SgemmFn.apply(P, P, 1.0 / self._t)
        norm_err = (sim.diagonal().detach() * self._t - 1).abs().max()
        assert norm_err.item() < 1e-4, "features need to be normalized first"
        sim_logits = sim - sim.max().detach()
        sim_exp = torch.exp(sim_logits)
        self.sim_exp, self.sim_logits, self.pos_mask, self.neg_mask = sim_exp, sim_logits, pos_mask, neg_mask
        pos_count = pos_mask.sum(1)
        denom = ((sim_exp * pos_mask).sum(1, keepdim=True) + (sim_exp * neg_mask).sum(1, keepdim=True))
        llh = sim_logits - torch.log(denom + 1e-16)
        with torch.no_grad():
            l_ij = -llh
            if self._weight_update == "hard":
                w = (l_ij <= self._gamma).float()
            else:
                w = torch.clamp(1 - l_ij / self._gamma, min=0)
            sp_mask = torch.max(w, 1 - pos_mask)
        self.sp_mask = sp_mask
        self.downgrade_ratio = torch.masked_select(sp_mask, pos_mask.bool()).mean().item()
        loss = -(((llh * sp_mask) * pos_mask).sum(1) / pos_count).mean()
        if self._correct_grad and self.downgrade_ratio > 0:
            loss = loss / self.downgrade_ratio
        if torch.isnan(loss):
            raise RuntimeError(loss)
        return loss
