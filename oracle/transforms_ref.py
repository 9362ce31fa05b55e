"""ORACLE — test infrastructure, not product code.

CPU restatement of the deterministic arithmetic of the viscy-transforms ops on the hot path
(/root/reference/packages/viscy-transforms/src/viscy_transforms/...), with the random
parameters *injected* (kornia / MONAI RNG streams are not reproducible here).
"""

from __future__ import annotations

import torch
from torch import Tensor


def _match(t: Tensor, target: Tensor) -> Tensor:
    """_normalize.py:22-24."""
    return t.reshape(t.shape + (1,) * (target.ndim - t.ndim))


def normalize_sampled(x: Tensor, subtrahend: Tensor, divisor: Tensor) -> Tensor:
    """NormalizeSampled.__call__, _normalize.py:72-80: (x - sub) / (div + 1e-8)."""
    return (x - _match(subtrahend, x)) / (_match(divisor, x) + 1e-8)


def minmax_sampled(x: Tensor, low: Tensor, high: Tensor) -> Tensor:
    """MinMaxSampled.__call__, _normalize.py:124-134."""
    lo, hi = _match(low, x), _match(high, x)
    x = x.clamp(lo, hi)
    return 2.0 * (x - lo) / (hi - lo + 1e-8) - 1.0


def scale_intensity(x: Tensor, factors: Tensor) -> Tensor:
    """BatchedRandScaleIntensity.__call__, _scale_intensity.py:59-77 with injected factors
    ((B,) or (B,C); zero where the transform is not applied): x * (1 + f)."""
    f = 1.0 + factors
    return x * f.view(*f.shape, *([1] * (x.ndim - f.ndim)))


def adjust_contrast(x: Tensor, gamma: Tensor, apply: Tensor, invert_image: bool = False) -> Tensor:
    """BatchedRandAdjustContrast.__call__, _adjust_contrast.py:54-86, which loops MONAI
    ``AdjustContrast(gamma)`` (monai 1.5.2 transforms/intensity/array.py) per sample:
    eps=1e-7; m=x.min(); r=x.max()-m; ((x-m)/(r+eps))**gamma * r + m  (retain_stats=False)."""
    out = torch.empty_like(x)
    for i in range(x.shape[0]):
        s = x[i]
        if bool(apply[i]):
            if invert_image:
                s = -s
            m = s.min()
            r = s.max() - m
            s = ((s - m) / (r + 1e-7)) ** float(gamma[i]) * r + m
            if invert_image:
                s = -s
        out[i] = s
    return out


def gaussian_noise(x: Tensor, noise: Tensor, std: Tensor, apply: Tensor, mean: float = 0.0) -> Tensor:
    """BatchedRandGaussianNoise.__call__, _noise.py:158-204 with injected field: one N(0,1)
    field shared across the batch (shape x.shape[1:]); selected samples get
    ``x + (mean + noise * std_b)`` (``addcmul`` then ``index_add``), others are untouched."""
    add = mean + noise.unsqueeze(0) * _match(std, x)
    return torch.where(_match(apply.bool(), x), x + add, x)
