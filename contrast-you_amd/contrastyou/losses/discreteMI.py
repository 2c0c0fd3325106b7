"""Discrete mutual-information (IIC) losses with the interface of
contrastyou/losses/discreteMI.py:90-170,201-261:

    IIDLoss(lamb)(x_out, x_tf_out)                 -> (loss, loss_no_lamb, p_i_j)   on [n,k] simplexes
    IIDSegmentationLoss(lamda, padding, eps, symmetric)(x_out, x_tf_out, mask=None) -> loss
                                                                                  on [n,k,H,W] maps
    compute_joint / compute_joint_2D / compute_joint_2D_with_padding_zeros

The k x k (or (2p+1)^2 x k x k) joint is a contraction over all n*H*W pixels -- HBM-bound, k = 20 --
done by the HIP joint kernels; the loss on the joint and its gradient come from one single-block
kernel (csrc/cy_mi.hip).  Inputs must be probability maps; the reference's `simplex()` assertion is a
host sync and is evaluated on demand by `validate()`.
"""
from __future__ import annotations

import sys
from typing import Tuple

import torch
from torch import Tensor, nn

from cyhip import ops
from cyhip.functions import IIDFn

__all__ = ["IIDLoss", "IIDSegmentationLoss", "compute_joint", "compute_joint_2D",
           "compute_joint_2D_with_padding_zeros"]


def _nhwc_f32(x: Tensor) -> Tensor:
    """[n,k,H,W] (any memory format) -> contiguous f32 [n,H,W,k] buffer (no copy for NHWC f32)"""
    x = x if x.dtype == torch.float32 else x.float()
    return x.permute(0, 2, 3, 1).contiguous()


class IIDLoss(nn.Module):

    def __init__(self, lamb: float = 1.0, eps: float = sys.float_info.epsilon):
        super().__init__()
        self.lamb, self.eps = float(lamb), float(eps)

    def forward(self, x_out: Tensor, x_tf_out: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        assert x_out.dim() == 2 and x_out.shape == x_tf_out.shape, (x_out.shape, x_tf_out.shape)
        ops.require_gpu(x_out, x_tf_out)
        n, k = x_out.shape
        a = x_out.float().contiguous().view(n, 1, 1, k)
        b = x_tf_out.float().contiguous().view(n, 1, 1, k)
        loss, loss_no_lamb, P = IIDFn.apply(a, b, 2, 0, True, self.lamb, 1e-10)  # the reference hard-codes 1e-10
        return loss, loss_no_lamb, P.view(k, k)


class IIDSegmentationLoss(nn.Module):

    def __init__(self, lamda=1.0, padding=0, eps: float = 1e-5, symmetric: bool = False) -> None:
        super().__init__()
        if padding < 0:
            raise ValueError(padding)
        self.lamda, self.padding, self._eps, self.symmetric = lamda, int(padding), eps, symmetric
        self._p_i_j = None

    def forward(self, x_out: Tensor, x_tf_out: Tensor, mask: Tensor = None) -> Tensor:
        assert x_out.shape == x_tf_out.shape and x_out.dim() == 4, (x_out.shape, x_tf_out.shape)
        ops.require_gpu(x_out, x_tf_out)
        if mask is not None:
            x_out, x_tf_out = x_out * mask, x_tf_out * mask
        a, b = _nhwc_f32(x_out), _nhwc_f32(x_tf_out)
        mode = 0 if self.padding == 0 else 1
        loss, _, P = IIDFn.apply(a, b, mode, self.padding, bool(self.symmetric), float(self.lamda), float(self._eps))
        self._p_i_j = P[0]
        return loss

    def get_joint_matrix(self):
        if self._p_i_j is None:
            raise RuntimeError()
        return self._p_i_j.detach().cpu().numpy()


@torch.no_grad()
def compute_joint(x_out: Tensor, x_tf_out: Tensor, symmetric=True) -> Tensor:
    """[n,k] x2 -> normalised (optionally symmetrised) k x k joint (discreteMI.py:201-222); no grad"""
    n, k = x_out.shape
    J = ops.joint_fwd(x_out.float().contiguous(), x_tf_out.float().contiguous(), n, 1, 1, k, 0, False)
    _, P, _ = ops.iid_loss(J, 2, bool(symmetric), 1.0, 1e-10, want_grad=False)
    return P.view(k, k)


@torch.no_grad()
def compute_joint_2D(x_out: Tensor, x_tf_out: Tensor, *, symmetric: bool = True, padding: int = 0) -> Tensor:
    """[n,k,H,W] x2 -> [T,T,k,k] displaced joint, T = 2*padding+1 (discreteMI.py:225-243); no grad"""
    n, k, H, W = x_out.shape
    J = ops.joint_fwd(_nhwc_f32(x_out), _nhwc_f32(x_tf_out), n, H, W, k, padding, False)
    _, P, _ = ops.iid_loss(J, 1, bool(symmetric), 1.0, 1e-5, want_grad=False)
    T = 2 * padding + 1
    return P.view(T, T, k, k)


@torch.no_grad()
def compute_joint_2D_with_padding_zeros(x_out: Tensor, x_tf_out: Tensor, *, symmetric: bool = True) -> Tensor:
    """[n,k,H,W] x2 -> [1,1,k,k] (discreteMI.py:246-261); no grad"""
    n, k, H, W = x_out.shape
    J = ops.joint_fwd(_nhwc_f32(x_out), _nhwc_f32(x_tf_out), n, H, W, k, 0, True)
    _, P, _ = ops.iid_loss(J, 0, bool(symmetric), 1.0, 1e-5, want_grad=False)
    return P.view(1, 1, k, k)
