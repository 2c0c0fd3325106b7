"""In-step affine augmentation on the device.

The reference wraps the third-party `rising` transforms (semi_seg/augment.py:286-311 with the
parameter ranges of semi_seg/epochers/epocher.py:226-238): scale U(0.8,1.3), rotation U(-45,45)
degrees, translation U(-0.1,0.1), nearest interpolation, a mirror along one random spatial dim
with p=0.9, and (image mode only) gamma U(0.5,2) applied BEFORE the geometry.  `rising` is not
vendored and unpinned, so its RNG consumption cannot be reproduced: this class draws the same
parameter families from its own seeded generator and resamples with the HIP nearest-neighbour
kernels (cyhip.functions.AffineFn).  What IS preserved is the contract the hooks rely on: the
same `seed` gives the same geometry for images, logits and feature maps of any resolution.
"""
from __future__ import annotations

import math
from contextlib import contextmanager
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from cyhip.functions import AffineFn


class AffineAugment:

    def __init__(self, *, scale=(0.8, 1.3), rotation=(-45.0, 45.0), translation=(-0.1, 0.1),
                 mirror_p: float = 0.9, gamma: Optional[Tuple[float, float]] = (0.5, 2.0)) -> None:
        self.scale, self.rotation, self.translation = scale, rotation, translation
        self.mirror_p, self.gamma = mirror_p, gamma
        self.enabled = True
        self._cache = {}

    @contextmanager
    def disabled(self):
        """identity transform inside the block (the reference's `disable_rising_augmentation`,
        semi_seg/epochers/pretrain.py:128-152, sets p=0 on every rising transform)"""
        prev, self.enabled = self.enabled, False
        try:
            yield self
        finally:
            self.enabled = prev

    def sample(self, n: int, seed: int):
        """host-side parameters for a batch of n: theta [n,2,3] (output->input, normalised coords,
        align_corners=False) and gamma [n]"""
        rs = np.random.RandomState(seed % (2 ** 32))
        theta = np.zeros((n, 2, 3), dtype=np.float32)
        sc = rs.uniform(*self.scale, size=n)
        rot = np.deg2rad(rs.uniform(*self.rotation, size=n))
        tr = rs.uniform(*self.translation, size=(n, 2))
        dims = rs.randint(0, 2, size=n)
        flip = rs.uniform(size=n) < self.mirror_p
        gam = rs.uniform(*self.gamma, size=n).astype(np.float32) if self.gamma else None
        for i in range(n):
            c, s = math.cos(rot[i]) / sc[i], math.sin(rot[i]) / sc[i]
            m = np.array([[c, -s, tr[i, 0]], [s, c, tr[i, 1]]], dtype=np.float32)
            if flip[i]:
                m[:, 1 - dims[i]] *= -1.0  # dim 0 = height (y column), dim 1 = width (x column)
            theta[i] = m
        return theta, gam

    def params(self, n: int, seed: int, device):
        key = (n, seed, str(device))
        hit = self._cache.get(key)
        if hit is None:
            theta, gam = self.sample(n, seed)
            from cyhip.ops import pinned
            hit = (pinned.upload(torch.from_numpy(theta).reshape(n, 6), device),
                   None if gam is None else pinned.upload(torch.from_numpy(gam), device))
            self._cache = {key: hit}  # one step's worth
        return hit

    def __call__(self, image: Tensor, *, mode: str, seed: int) -> Tensor:
        assert mode in {"image", "feature"}, f"`mode` must be in `image` or `feature`, given {mode}."
        if not self.enabled:
            return image.float() if mode == "image" else image
        theta, gam = self.params(image.shape[0], seed, image.device)
        if mode == "image":
            return AffineFn.apply(image.float(), theta, gam)
        return AffineFn.apply(image, theta, None)


# name kept so that code written against the reference keeps importing
RisingWrapper = AffineAugment
