"""ORACLE (test infrastructure, not product code): one SemiSupervisedEpocher training step on the
CPU, composed from the oracle leaves in the reference's order
(semi_seg/epochers/epocher.py:297-360 `_batch_update` / `_forward_pass`, and
semi_seg/hooks/infonce.py:222-245 `_INFONCEEpochHook._call_implementation`):

    two-stage forward:  logits_l = net(labeled);  logits_u, logits_utf = net(cat[unl, unl_tf])
    sup   = KL_div(softmax(logits_l), one_hot(target))
    feats = cat(Conv5 outputs of both calls)[-2*n_unl:] -> (f_u, f_utf)
    z     = projector(cat[affine(f_u), f_utf]) -> (z1, z2)
    reg   = weight * SupConLoss1(z1, z2, target=labels)
    total = sup + reg ; backward ; RAdam step

The affine geometry is an explicit input (theta), because the reference delegates it to the
un-vendored `rising` package (parity unpinned there; see oracle.losses.affine_nearest).
Used by tests (composed-step parity of the HIP path) and by bench.py's cpu_baseline leg.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import losses as ol
from . import unet as ou


def semi_step(sd: Dict[str, Tensor], psd: Dict[str, Tensor], *, labeled_image: Tensor, labeled_target: Tensor,
              unlabeled_image: Tensor, unlabeled_image_tf: Tensor, theta: Tensor, labels: List[int],
              momentum: float = 0.01, weight: float = 1.0, two_stage: bool = True,
              feature_name: str = "Conv5", round_dtype: Optional[torch.dtype] = None,
              force: Optional[dict] = None):
    """forward + losses of one step; `sd`/`psd` tensors that require grad receive .grad after
    `total.backward()`.  Returns dict(total, sup, reg, label_logits, unlabeled_logits_tf).
    `force`: the device's own per-layer records of its passes in call order (oracle.unet.unet_forward)."""
    n_l, n_unl = labeled_image.shape[0], unlabeled_image.shape[0]
    feats_a, feats_b = {}, {}
    if two_stage:
        label_logits = ou.unet_forward(sd, labeled_image, training=True, momentum=momentum, feats=feats_a,
                                       round_dtype=round_dtype, force=force)
        both = ou.unet_forward(sd, torch.cat([unlabeled_image, unlabeled_image_tf], 0), training=True,
                               momentum=momentum, feats=feats_b, round_dtype=round_dtype, force=force)
        unl_logits, unl_tf_logits = torch.split(both, [n_unl, n_unl], 0)
        collected = torch.cat([feats_a[feature_name], feats_b[feature_name]], 0)
    else:
        allv = ou.unet_forward(sd, torch.cat([labeled_image, unlabeled_image, unlabeled_image_tf], 0),
                               training=True, momentum=momentum, feats=feats_a, round_dtype=round_dtype, force=force)
        label_logits, unl_logits, unl_tf_logits = torch.split(allv, [n_l, n_unl, n_unl], 0)
        collected = feats_a[feature_name]
    unl_logits_tf = ol.affine_nearest(unl_logits, theta)
    sup = ol.sup_loss(label_logits, labeled_target.squeeze(1))
    f_u, f_utf = torch.chunk(collected[-2 * n_unl:], 2, 0)
    z = ol.projection_head(psd, torch.cat([ol.affine_nearest(f_u, theta), f_utf], 0))
    z1, z2 = torch.chunk(z, 2, 0)
    reg = ol.supcon_loss(z1, z2, target=labels) * weight
    return {"total": sup + reg, "sup": sup, "reg": reg, "label_logits": label_logits,
            "unlabeled_logits_tf": unl_logits_tf, "unlabeled_tf_logits": unl_tf_logits}


def synthetic_batch(n_l: int, n_unl: int, hw: int, num_classes: int, seed: int = 1234):
    """ACDC-shaped synthetic tensors: U[0,1) images, blob-free random labels,
    partitions cycling "0","1","2", scan ids patientXXX_YY."""
    g = torch.Generator().manual_seed(seed)
    img = lambda n: torch.rand(n, 1, hw, hw, generator=g)  # noqa: E731
    batch = {
        "labeled_image": img(n_l), "labeled_target": torch.randint(0, num_classes, (n_l, 1, hw, hw), generator=g),
        "unlabeled_image": img(n_unl), "unlabeled_image_cf": img(n_unl),
        "partition": [str(i % 3) for i in range(n_unl)],
        "scan": [f"patient{i // 3:03d}_{i % 2:02d}" for i in range(n_unl)],
        "labeled_scan": [f"patient{100 + i // 3:03d}_{i % 2:02d}" for i in range(n_l)],
    }
    return batch


def radam_step(params: List[Tensor], state: dict, lr: float, weight_decay: float, betas=(0.9, 0.999),
               eps: float = 1e-8) -> None:
    """torch.optim.RAdam semantics (contrastyou/trainer/base.py:66-75) via the stock optimizer"""
    if "opt" not in state:
        state["opt"] = torch.optim.RAdam(params, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
    state["opt"].step()
