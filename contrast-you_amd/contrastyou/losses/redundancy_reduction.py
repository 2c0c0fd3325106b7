"""`RedundancyCriterion` (contrastyou/losses/redundancy_reduction.py:12-55; adapted there from Barlow Twins,
arXiv 2103.03230): pull the joint of two cluster-probability maps towards the diagonal.

    p_ij   = compute_joint_2D_with_padding_zeros(x_out, x_tf_out, symmetric)     (k x k, sums to 1)
    target = alpha * eye(k) / k + (1 - alpha) * p_ij
    loss   = -sum(target * log(p_ij + eps)) + lamda * sum(p_ij * (log(p_j + eps) + log(p_i + eps)))

The contraction over all n*H*W pixels is the HIP joint kernel (cyhip.functions.JointFn); the arithmetic on the
k x k result (k <= 40) is plain autograd.  NB, as in the reference, `target` is not detached: the gradient flows
through its p_ij term too.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from cyhip import ops
from cyhip.functions import JointFn

__all__ = ["RedundancyCriterion"]


class RedundancyCriterion(nn.Module):

    def __init__(self, *, eps: float = 1e-5, symmetric: bool = True, lamda: float = 1, alpha: float) -> None:
        super().__init__()
        self._eps, self.symmetric, self.lamda, self.alpha = eps, symmetric, lamda, alpha
        self._p_i_j = None

    def forward(self, x_out: Tensor, x_tf_out: Tensor) -> Tensor:
        assert x_out.shape == x_tf_out.shape and x_out.dim() == 4, (x_out.shape, x_tf_out.shape)
        ops.require_gpu(x_out, x_tf_out)
        k = x_out.shape[1]
        a = x_out.float().permute(0, 2, 3, 1).contiguous()
        b = x_tf_out.float().permute(0, 2, 3, 1).contiguous()
        p = JointFn.apply(a, b)
        if self.symmetric:
            p = (p + p.t()) / 2.0
        self._p_i_j = p
        eye = torch.eye(k, device=p.device, dtype=p.dtype)
        target = (eye / k) * self.alpha + p * (1 - self.alpha)
        p_i = p.sum(dim=1).view(k, 1).expand(k, k)
        p_j = p.sum(dim=0).view(1, k).expand(k, k)
        constrained = (-p * (-self.lamda * torch.log(p_j + self._eps) - self.lamda * torch.log(p_i + self._eps))).sum()
        pseudo_loss = -(target * (p + self._eps).log()).sum()
        return pseudo_loss + constrained

    def kl_criterion(self, dist: Tensor, prior: Tensor):
        return -(prior * (dist + self._eps).log() + (1 - prior) * (1 - dist + self._eps).log()).mean()

    def get_joint_matrix(self):
        if self._p_i_j is None:
            raise RuntimeError()
        return self._p_i_j.detach().cpu().numpy()

    def set_ratio(self, alpha: float):
        """0: entropy minimisation, 1: Barlow-Twins-style diagonal target"""
        assert 0 <= alpha <= 1, alpha
        self.alpha = alpha
