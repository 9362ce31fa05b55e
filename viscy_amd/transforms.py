"""viscy-transforms surface used by the UNeXt2 recipes, on MI355X.

Same class names / constructor keywords / dict-in-dict-out call convention as
``viscy_transforms`` (/root/reference/packages/viscy-transforms/src/viscy_transforms):
``NormalizeSampled``, ``MinMaxSampled`` (_normalize.py:27-134), ``BatchedRandScaleIntensityd``
(_scale_intensity.py), ``BatchedRandAdjustContrastd`` (_adjust_contrast.py),
``BatchedRandGaussianNoised`` (_noise.py), ``BatchedRandFlipd`` (_flip.py),
``BatchedCenterSpatialCropd`` (_crop.py:164-213).

Tensors on the GPU go through the fused HIP kernels of csrc/transforms.hip (one pass for the whole
contrast → scale → noise chain).  ``NormalizeSampled`` / ``MinMaxSampled`` are also applied by the
reference inside DataLoader *worker processes* on CPU tensors (hcs.py:783); workers must never touch
the GPU library, so CPU inputs use the identical two-line arithmetic in torch — that is host-side
data preparation, not a fallback for a GPU kernel.  Random parameters are sampled on the host with
torch's generator (RNG streams of kornia / MONAI cannot be reproduced); every transform accepts
``params=`` to inject them, which is how the parity tests drive it.
"""

from __future__ import annotations

import math

from typing import Iterable, Sequence

import torch
from torch import Tensor

from . import _lib as L
from ._lib import check, lib, ptr, stream

_DATA_RANGE_KEYS = {"min_max": ("min", "max"), "p1_p99": ("p1", "p99"), "p5_p95": ("p5", "p95")}


def _keys(keys) -> tuple[str, ...]:
    return (keys,) if isinstance(keys, str) else tuple(keys)


def _stat(t, B: int, dev) -> Tensor:
    t = torch.as_tensor(t, dtype=torch.float32)
    if t.ndim == 0:
        t = t.expand(B)
    return t.reshape(B).to(dev, non_blocking=True).contiguous()


def _match(t: Tensor, target: Tensor) -> Tensor:
    return t.reshape(t.shape + (1,) * (target.ndim - t.ndim)).to(device=target.device)


def _gpu_ok(x: Tensor) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and (x[0].numel() % 4 == 0)


class NormalizeSampled:
    """(x - subtrahend) / (divisor + 1e-8) with precomputed statistics from ``sample["norm_meta"]``."""

    is_spatial = False

    def __init__(self, keys, level, subtrahend="mean", divisor="std", remove_meta: bool = False):
        self.keys, self.level = _keys(keys), level
        self.subtrahend, self.divisor, self.remove_meta = subtrahend, divisor, remove_meta

    def __call__(self, sample: dict) -> dict:
        for key in self.keys:
            meta = sample["norm_meta"][key][self.level]
            x = sample[key]
            sub, div = meta[self.subtrahend], meta[self.divisor]
            if _gpu_ok(x) and x.ndim >= 2:
                B = x.shape[0]
                y = torch.empty_like(x)
                sub_d, div_d = _stat(sub, B, x.device), _stat(div, B, x.device)  # keep alive across the launch
                check(lib().vsx_normalize(ptr(x), ptr(y), ptr(sub_d), ptr(div_d), B, x[0].numel(), stream()), "normalize")
                sample[key] = y
            else:  # DataLoader worker / CPU tensor
                sample[key] = (x - _match(torch.as_tensor(sub), x)) / (_match(torch.as_tensor(div), x) + 1e-8)
        if self.remove_meta:
            sample.pop("norm_meta")
        return sample


def normalize_stacked(x: Tensor, sub: Tensor, div: Tensor) -> Tensor:
    """(x - sub[b, c]) / (div[b, c] + 1e-8) for a stacked (B, C, ...) batch on the device — the data module's
    on-device form of ``NormalizeSampled`` (identity channels: sub 0, div 1)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x[0, 0].numel() % 4 == 0):
        raise RuntimeError("normalize_stacked needs a contiguous float32 (B,C,...) batch on the HIP device (no CPU fallback)")
    B, C = x.shape[:2]
    y = torch.empty_like(x)
    sub_d, div_d = _stat(sub, B * C, x.device), _stat(div, B * C, x.device)
    check(lib().vsx_normalize(ptr(x), ptr(y), ptr(sub_d), ptr(div_d), B * C, x[0, 0].numel(), stream()), "normalize")
    return y


class MinMaxSampled:
    """clamp to [low, high] then rescale to [-1, 1]."""

    is_spatial = False

    def __init__(self, keys, level, data_range="p1_p99", remove_meta: bool = False):
        if data_range not in _DATA_RANGE_KEYS:
            raise ValueError(f"Invalid data_range: {data_range}")
        self.keys, self.level, self.remove_meta = _keys(keys), level, remove_meta
        self._low_key, self._high_key = _DATA_RANGE_KEYS[data_range]

    def __call__(self, sample: dict) -> dict:
        for key in self.keys:
            meta = sample["norm_meta"][key][self.level]
            x = sample[key]
            lo, hi = meta[self._low_key], meta[self._high_key]
            if _gpu_ok(x) and x.ndim >= 2:
                B = x.shape[0]
                y = torch.empty_like(x)
                lo_d, hi_d = _stat(lo, B, x.device), _stat(hi, B, x.device)
                check(lib().vsx_minmax_norm(ptr(x), ptr(y), ptr(lo_d), ptr(hi_d), B, x[0].numel(), stream()), "minmax_norm")
                sample[key] = y
            else:
                lo_t, hi_t = _match(torch.as_tensor(lo), x), _match(torch.as_tensor(hi), x)
                xc = x.clamp(lo_t, hi_t)
                sample[key] = 2.0 * (xc - lo_t) / (hi_t - lo_t + 1e-8) - 1.0
        if self.remove_meta:
            sample.pop("norm_meta")
        return sample


def _sample_mean_std(x: Tensor) -> tuple[Tensor, Tensor]:
    """per-sample mean and unbiased std (``Tensor.mean()`` / ``Tensor.std()`` of MONAI's AdjustContrast) from one pass of
    double sums (csrc/transforms.hip::sample_moments_kernel); (B,) float64 on the device."""
    B, per = x.shape[0], x[0].numel()
    sums = torch.zeros((B, 2), dtype=torch.float64, device=x.device)
    check(lib().vsx_sample_moments(ptr(x), ptr(sums), B, per, stream()), "sample_moments")
    mean = sums[:, 0] / per
    var = (sums[:, 1] - per * mean * mean).clamp_min(0.0) / max(per - 1, 1)
    return mean, var.sqrt()


def intensity_augment(x: Tensor, *, gamma: Tensor | None = None, factor: Tensor | None = None,
                      noise: Tensor | None = None, noise_std: Tensor | None = None, noise_mean: float = 0.0,
                      invert_image: bool = False, retain_stats: bool = False, _invert: int | None = None) -> Tensor:
    """Fused contrast → scale → noise pass on a (B, ...) fp32 GPU batch (csrc/transforms.hip).
    gamma[b] <= 0: no contrast change; factor[b] = 0: no scaling; noise_std[b] < 0: no noise.
    ``factor`` may be (B,) or (B, C) (``channel_wise=True``: one factor per channel, _scale_intensity.py:46-55).
    ``invert_image`` / ``retain_stats``: the options of MONAI's ``AdjustContrast`` (forwarded at _adjust_contrast.py:76-80):
    the curve runs on -x and is negated back; mean and std of the (inverted) sample are restored after the curve."""
    if not _gpu_ok(x):
        raise RuntimeError("viscy_amd GPU augmentations need a contiguous float32 batch on the HIP device (no CPU fallback)")
    B, per = x.shape[0], x[0].numel()
    dev = x.device
    if factor is not None and factor.ndim == 2:
        # per-channel factors: the same pass over a (B*C, ...) view (contrast and noise are per sample: not combined with it)
        if gamma is not None or noise is not None:
            raise ValueError("per-channel factors go in a pass of their own")
        if x.ndim < 3 or factor.shape not in ((B, x.shape[1]), (B, 1)):
            raise RuntimeError(f"factors of shape {tuple(factor.shape)} do not broadcast over a batch of shape {tuple(x.shape)}")
        f = factor.expand(B, x.shape[1]).reshape(-1)
        return intensity_augment(x.view(B * x.shape[1], *x.shape[2:]), factor=f).view(x.shape)
    if retain_stats and gamma is not None:
        if factor is not None or noise is not None:
            raise ValueError("retain_stats is an option of the contrast stage alone")
        # MONAI: [invert] → mean / std → curve → (ret - mean(ret)) / (std(ret) + 1e-8) * std + mean → [invert back]
        sel = gamma.to(dev) > 0
        m0, s0 = _sample_mean_std(x)
        if invert_image:
            m0 = -m0
        ret = intensity_augment(x, gamma=gamma, _invert=1 if invert_image else 0)
        m1, s1 = _sample_mean_std(ret)
        div = (s1 + 1e-8) / s0
        sub = m1 - m0 * div
        # a constant sample (blank / zero-padded patch): s0 = 0 makes div inf and sub NaN, and one such sample would poison the
        # step.  MONAI returns the constant there ((ret - mean) = 0 is divided by 0 + 1e-8, then * 0 + mn): div = 1 and
        # sub = m1 - m0 give ret - m1 + m0 = m0 (the curve maps a constant onto itself, and its double mean is exact)
        flat = s0 == 0
        div = torch.where(flat, torch.ones_like(div), div)
        sub = torch.where(flat, m1 - m0, sub)
        if invert_image:
            div = -div
        one, zero = torch.ones_like(div), torch.zeros_like(sub)
        div_f = (torch.where(sel, div, one) - 1e-8).float()  # vsx_normalize divides by (div + 1e-8), NormalizeSampled's form
        sub_f = torch.where(sel, sub, zero).float()
        y = torch.empty_like(x)
        check(lib().vsx_normalize(ptr(ret), ptr(y), ptr(sub_f), ptr(div_f), B, per, stream()), "normalize")
        return y
    mn = mx = None
    if gamma is not None:
        mm = torch.empty(2, B, dtype=torch.float32, device=dev)
        mm[0].fill_(float("inf"))
        mm[1].fill_(float("-inf"))
        check(lib().vsx_sample_minmax(ptr(x), ptr(mm[0]), ptr(mm[1]), B, per, stream()), "sample_minmax")
        mn, mx = mm[0], mm[1]
    y = torch.empty_like(x)
    f32 = lambda t: None if t is None else t.to(dev, torch.float32).contiguous()  # noqa: E731
    g_d, f_d, n_d, s_d = f32(gamma), f32(factor), f32(noise), f32(noise_std)  # named: ptr() does not keep a tensor alive
    check(lib().vsx_intensity_aug(ptr(x), ptr(y), ptr(mn), ptr(mx), ptr(g_d), ptr(f_d), ptr(n_d), ptr(s_d),
                                  float(noise_mean), _invert if _invert is not None else (3 if invert_image else 0), B, per,
                                  stream()), "intensity_aug")
    return y


class _BatchedRand:
    is_spatial = False  # the spatial subclasses (flip, affine, crops) say so: foreground masks follow them and only them

    def __init__(self, keys, prob: float):
        self.keys, self.prob = _keys(keys), prob
        self.generator: torch.Generator | None = None

    def _rand(self, n):
        return torch.rand(n, generator=self.generator)


class BatchedRandScaleIntensityd(_BatchedRand):
    """x * (1 + f_b), f_b ~ U(factors) for selected samples (same factor for every key)."""

    def __init__(self, keys, factors=0.1, prob: float = 0.1, channel_wise: bool = False, allow_missing_keys: bool = False):
        super().__init__(keys, prob)
        self.channel_wise = channel_wise
        self.range = (-abs(factors), abs(factors)) if isinstance(factors, (int, float)) else (min(factors), max(factors))

    def randomize(self, B: int, C: int | None = None) -> Tensor:
        """the reference's draw order (_scale_intensity.py:42-50): selection first, then one factor per sample — or, with
        ``channel_wise``, per (sample, channel) of the FIRST key; unselected samples get 0"""
        do = self._rand(B) < self.prob
        shape = (B, C) if self.channel_wise and C is not None else (B,)
        f = torch.empty(shape).uniform_(*self.range, generator=self.generator)
        f[~do] = 0.0
        return f

    def __call__(self, sample: dict, params: Tensor | None = None) -> dict:
        first = sample[self.keys[0]]
        f = params if params is not None else self.randomize(first.shape[0], first.shape[1] if first.ndim > 2 else None)
        for k in self.keys:
            sample[k] = intensity_augment(sample[k], factor=f)
        return sample


class BatchedRandAdjustContrastd(_BatchedRand):
    """per-sample gamma contrast (MONAI AdjustContrast), gamma_b ~ U(gamma) for selected samples."""

    def __init__(self, keys, gamma=(0.5, 4.5), prob: float = 0.1, invert_image: bool = False, retain_stats: bool = False,
                 allow_missing_keys: bool = False):
        super().__init__(keys, prob)
        self.invert_image, self.retain_stats = invert_image, retain_stats
        self.gamma_range = (gamma, gamma) if isinstance(gamma, (int, float)) else (min(gamma), max(gamma))
        if self.gamma_range[0] <= 0.0:
            raise ValueError("Gamma must be a positive value.")

    def randomize(self, B: int) -> Tensor:
        g = torch.empty(B).uniform_(*self.gamma_range, generator=self.generator)
        g[~(self._rand(B) < self.prob)] = 0.0  # 0 = not selected
        return g

    def __call__(self, sample: dict, params: Tensor | None = None) -> dict:
        g = params if params is not None else self.randomize(sample[self.keys[0]].shape[0])
        for k in self.keys:
            sample[k] = intensity_augment(sample[k], gamma=g, invert_image=self.invert_image, retain_stats=self.retain_stats)
        return sample


class BatchedRandGaussianNoised(_BatchedRand):
    """one N(0,1) field shared by the batch, scaled by a per-sample std (selected samples only)."""

    def __init__(self, keys, prob: float = 0.1, mean: float = 0.0, std: float = 0.1, sample_std: bool = True,
                 allow_missing_keys: bool = False, dtype=None):
        super().__init__(keys, prob)
        self.mean, self.std, self.sample_std = mean, std, sample_std

    def randomize(self, x: Tensor):
        B = x.shape[0]
        do = self._rand(B) < self.prob
        std = self._rand(B) * self.std if self.sample_std else torch.full((B,), float(self.std))
        std[~do] = -1.0  # negative = not selected
        noise = torch.randn(x.shape[1:], device=x.device, dtype=torch.float32)
        return noise, std

    def __call__(self, sample: dict, params=None) -> dict:
        for k in self.keys:
            noise, std = params if params is not None else self.randomize(sample[k])
            sample[k] = intensity_augment(sample[k], noise=noise, noise_std=std, noise_mean=self.mean)
        return sample


class BatchedRandFlipd(_BatchedRand):
    """per-sample random flips along the given spatial axes (pure data movement: torch.flip)."""

    is_spatial = True

    def __init__(self, keys, spatial_axes: Sequence[int] = (0, 1, 2), prob: float = 0.5, allow_missing_keys: bool = False):
        super().__init__(keys, prob)
        self.spatial_axes = tuple(spatial_axes)
        self.allow_missing_keys = allow_missing_keys

    def __call__(self, sample: dict, params: Tensor | None = None) -> dict:
        first = next((k for k in self.keys if k in sample), None)
        if first is None:
            return sample
        B = sample[first].shape[0]
        flips = params if params is not None else (torch.rand(B, len(self.spatial_axes), generator=self.generator) < self.prob)
        for k in self.keys:
            if k not in sample:
                if self.allow_missing_keys:
                    continue
                raise KeyError(k)
            x = sample[k]
            out = x.clone()
            for b in range(B):
                dims = [a + 1 for a, f in zip(self.spatial_axes, flips[b]) if bool(f)]  # +1: channel dim of x[b]
                if dims:
                    out[b] = torch.flip(x[b], dims)
            sample[k] = out
        return sample


class BatchedCenterSpatialCropd:
    """centre crop of the trailing spatial dims to ``roi_size`` (slicing only)."""

    is_spatial = True

    def __init__(self, keys, roi_size: Sequence[int], allow_missing_keys: bool = False):
        self.keys, self.roi_size = _keys(keys), tuple(roi_size)
        self.allow_missing_keys = allow_missing_keys

    def __call__(self, sample: dict) -> dict:
        for k in self.keys:
            if k not in sample:
                if self.allow_missing_keys:
                    continue
                raise KeyError(k)
            x = sample[k]
            sl = [slice(None)] * x.ndim
            for d, size in zip(range(x.ndim - len(self.roi_size), x.ndim), self.roi_size):
                start = (x.shape[d] - size) // 2
                sl[d] = slice(start, start + size)
            sample[k] = x[tuple(sl)].contiguous()
        return sample


# ------------------------------------------------------------------------------------------------
# K22 BatchedRandGaussianSmooth / K18 BatchedRandAffined
# ------------------------------------------------------------------------------------------------
def _erf_taps(kernel_size: int, sigma: Tensor) -> Tensor:
    """pixel-integrated Gaussian taps (kornia get_gaussian_erf_kernel1d), σ = 0 → identity; (B, k) on the host."""
    r = kernel_size // 2
    i = torch.arange(-r, r + 1, dtype=torch.float64).view(1, -1)
    sg = sigma.double().abs().view(-1, 1)
    t = torch.where(sg > 0, 0.7071067811865476 / sg.clamp_min(1e-30), torch.full_like(sg, float("inf")))
    g = 0.5 * (torch.erf((i + 0.5) * t) - torch.erf((i - 0.5) * t))
    g = torch.nan_to_num(g, nan=0.0).clamp_min(0)
    return (g / g.sum(-1, keepdim=True)).float()


def gaussian_smooth(x: Tensor, sigma_zyx: Tensor, apply: Tensor, truncated: float = 4.0) -> Tensor:
    """separable 3-D Gaussian with per-sample sigma (B, 3) in (Z, Y, X) order, zero border (csrc/transforms.hip)."""
    if not _gpu_ok(x) or x.ndim != 5:
        raise RuntimeError("viscy_amd GPU augmentations need a contiguous float32 (B,C,Z,Y,X) batch on the HIP device (no CPU fallback)")
    B, C, D, H, W = x.shape
    sel = apply.bool().cpu()
    if not sel.any():
        return x
    sg = torch.where(sel.view(-1, 1), sigma_zyx.float().cpu(), torch.zeros(B, 3))
    cur = x
    per = C * D * H * W
    for axis, (stride, L) in enumerate(((H * W, D), (W, H), (1, W))):
        s = sg[:, axis]
        if not (s[sel] > 0).any():
            continue
        tail = int(max(float(s[sel].max()) * truncated, 0.5) + 0.5)
        k = 2 * tail + 1
        taps = _erf_taps(k, s).to(x.device).contiguous()  # unselected samples (σ = 0) get the identity kernel
        out = torch.empty_like(cur)
        check(lib().vsx_conv1d_axis(ptr(cur), ptr(out), ptr(taps), k, B, per, stride, L, stream()), "conv1d_axis")
        cur = out
    return cur


class BatchedRandGaussianSmoothd(_BatchedRand):
    def __init__(self, keys, sigma_x=(0.25, 1.5), sigma_y=(0.25, 1.5), sigma_z=(0.25, 1.5), truncated: float = 4.0,
                 approx: str = "erf", prob: float = 0.1, allow_missing_keys: bool = False):
        super().__init__(keys, prob)
        self.ranges = [self._rng(sigma_z), self._rng(sigma_y), self._rng(sigma_x)]  # (Z, Y, X)
        self.truncated = truncated

    @staticmethod
    def _rng(s):
        if isinstance(s, (int, float)):
            return (float(s), float(s))
        if len(s) != 2:
            raise ValueError(f"sigma must be float or tuple of 2 values, got {s}")
        return (float(s[0]), float(s[1]))

    def randomize(self, B: int):
        do = self._rand(B) < self.prob
        sig = torch.stack([self._rand(B) * (hi - lo) + lo for lo, hi in self.ranges], dim=1)
        return sig, do

    def __call__(self, sample: dict, params=None) -> dict:
        sig, do = params if params is not None else self.randomize(sample[self.keys[0]].shape[0])
        for k in self.keys:
            sample[k] = gaussian_smooth(sample[k], sig, do, self.truncated)
        return sample


def _center_window(shape, roi_size):
    """(z0, y0, x0, Do, Ho, Wo) of BatchedCenterSpatialCrop on a (.., D, H, W) frame (start = (dim - size) // 2)."""
    dims = list(shape[-3:])
    size = [min(int(s), d) if int(s) > 0 else d for s, d in zip(roi_size, dims)]
    return tuple((d - s) // 2 for d, s in zip(dims, size)) + tuple(size)


_PADDING_MODES = {"zeros": 0, "border": 1, "reflection": 2}


def kornia_sampling_matrix(Minv: Tensor, shape_dhw, align_corners: bool) -> Tensor:
    """The voxel mapping kornia's ``warp_affine3d`` REALLY applies for an output→input voxel matrix ``Minv`` (B,3,4; x, y, z order).
    kornia normalises the matrix with (size - 1) denominators whatever ``align_corners`` is, then calls ``affine_grid`` /
    ``grid_sample`` with the flag: with ``align_corners=True`` the two conventions agree and output voxel i samples ``Minv·i``; with
    ``align_corners=False`` — the default of ``kornia.augmentation.RandomAffine3D``, which the reference forwards untouched
    (_affine.py:33-47) — output voxel i sits at (i + 0.5)·a in the matrix' frame, a = (size - 1) / size per axis, and the result
    is mapped back by 1 / a and - 0.5:  x' = Da^-1 (R·Da·(i + 0.5) + t) - 0.5  (third-party behaviour, restated)."""
    if align_corners:
        return Minv
    D, H, W = (int(v) for v in shape_dhw)
    a = torch.tensor([(n - 1) / n if n > 1 else 1.0 for n in (W, H, D)], dtype=torch.float64, device=Minv.device)
    M = Minv.to(torch.float64)
    R, t = M[:, :, :3], M[:, :, 3]
    Rp = R * a.view(1, 1, 3) / a.view(1, 3, 1)
    tp = (0.5 * (R * a.view(1, 1, 3)).sum(-1) + t) / a.view(1, 3) - 0.5
    return torch.cat([Rp, tp.unsqueeze(-1)], dim=-1).to(Minv.dtype)


def warp_affine3d(x: Tensor, Minv: Tensor, mode: str = "bilinear", window=None, padding_mode: str = "zeros",
                  align_corners: bool = False) -> Tensor:
    """resample (B,C,D,H,W) with the output→input voxel matrices Minv (B,3,4) (csrc/transforms.hip); ``padding_mode`` as in
    the reference (_affine.py:102-108): "zeros", "border" (edge voxels replicated) or "reflection"; ``align_corners`` as kornia's
    ``warp_affine3d`` receives it from ``RandomAffine3D`` (default False, see ``kornia_sampling_matrix``).
    ``window = (z0, y0, x0, Do, Ho, Wo)`` produces only that region of the output frame (warp + crop in one pass)."""
    if padding_mode not in _PADDING_MODES:
        raise ValueError(f"padding_mode must be one of {sorted(_PADDING_MODES)}, got {padding_mode!r}")
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 5):
        raise RuntimeError("viscy_amd GPU augmentations need a contiguous float32 (B,C,Z,Y,X) batch on the HIP device (no CPU fallback)")
    B, C, D, H, W = x.shape
    m = kornia_sampling_matrix(Minv.to(x.device), (D, H, W), align_corners).to(torch.float32).contiguous()
    pad_code = 3 if (padding_mode == "reflection" and not align_corners) else _PADDING_MODES[padding_mode]
    z0, y0, x0, Do, Ho, Wo = window if window is not None else (0, 0, 0, D, H, W)
    y = torch.empty((B, C, Do, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib().vsx_warp_affine3d_roi(ptr(x), ptr(y), ptr(m), B, C, D, H, W, z0, y0, x0, Do, Ho, Wo,
                                      int(mode == "nearest") | (pad_code << 1), stream()), "warp_affine3d")
    return y


def _axis_angle_matrix(rot_rad: Tensor) -> Tensor:
    """rotation vector (B, 3) [radians, axis * angle] -> (B, 3, 3), Rodrigues (kornia ``axis_angle_to_rotation_matrix``)"""
    th = rot_rad.norm(dim=1, keepdim=True).clamp_min(1e-12)
    k = rot_rad / th
    K = torch.zeros(rot_rad.shape[0], 3, 3, dtype=rot_rad.dtype)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    th = th.view(-1, 1, 1)
    return torch.eye(3, dtype=rot_rad.dtype) + th.sin() * K + (1 - th.cos()) * (K @ K)


def kornia_affine_matrix3d(angles_xyz_deg: Tensor, scale_xyz: Tensor, shears_deg: Tensor, translations_xyz: Tensor,
                           shape_dhw) -> Tensor:
    """Forward (input -> output) voxel matrix (B, 4, 4) of kornia 0.8.3 ``get_affine_matrix3d`` as ``RandomAffine3D`` calls it
    (restated from the published source — kornia is not under /root/reference: parity unpinned): rotation vector
    ``-angles`` about the volume centre ((W-1)/2, (H-1)/2, (D-1)/2), per-axis scale, translation, then the six shear facets
    (sxy, sxz, syx, syz, szx, szy; tangent of degrees) about the same centre.  All in kornia's (x, y, z) order, float64."""
    D, H, W = shape_dhw
    B = angles_xyz_deg.shape[0]
    f = torch.float64
    ctr = torch.tensor([(W - 1) / 2.0, (H - 1) / 2.0, (D - 1) / 2.0], dtype=f)
    R = _axis_angle_matrix(torch.deg2rad(-angles_xyz_deg.to(f))) @ torch.diag_embed(scale_xyz.to(f))
    P = torch.eye(4, dtype=f).repeat(B, 1, 1)
    P[:, :3, :3] = R
    frm, to = torch.eye(4, dtype=f).repeat(B, 1, 1), torch.eye(4, dtype=f).repeat(B, 1, 1)
    frm[:, :3, 3], to[:, :3, 3] = ctr, -ctr
    M = frm @ P @ to
    M[:, :3, 3] += translations_xyz.to(f)
    t = torch.tan(torch.deg2rad(shears_deg.to(f)))
    sxy, sxz, syx, syz, szx, szy = t.unbind(1)
    one = torch.ones_like(sxy)
    m00, m10, m20 = one, sxy, sxz
    m01, m11, m21 = syx, sxy * syx + one, sxz * syx + syz
    m02 = syx * szy + szx
    m12 = sxy * szx + szy * m11
    m22 = sxz * szx + szy * m21 + one
    x, y, z = ctr
    m03 = m01 * y + m02 * z
    m13 = m10 * x + m11 * y + m12 * z - y
    m23 = m20 * x + m21 * y + m22 * z - z
    S = torch.eye(4, dtype=f).repeat(B, 1, 1)
    S[:, 0, :] = torch.stack([m00, -m01, -m02, m03], 1)
    S[:, 1, :] = torch.stack([-m10, m11, -m12, m13], 1)
    S[:, 2, :] = torch.stack([-m20, -m21, m22, m23], 1)
    return M @ S


class BatchedRandAffined(_BatchedRand):
    """``viscy_transforms.BatchedRandAffined`` (_affine.py:107-393): one random 3-D affine per sample, the same matrix for
    every key, trilinear resampling with zero / border / reflection padding (``vsx_warp_affine3d``).  ``align_corners`` (class
    attribute, False) is what kornia's ``RandomAffine3D`` hands to ``warp_affine3d`` by default; the reference never sets it
    (_affine.py:33-47, :180-200).  Arguments as in the reference:

    * ``rotate_range`` — radians per axis in (Z, Y, X) order, a value ``v`` meaning ``(-v, v)`` (or explicit ``(lo, hi)``);
    * ``shear_range`` — degrees: ``(min, max)`` for all six facets, six ``(min, max)`` pairs, or MONAI's three-value
      shorthand ``[s_zy, s_zx, s_yz]`` (-> facets szy, szx, syz = ±value); ``scale_z_shear`` multiplies the Z-related facets
      by ``depth / max(Y, X)`` (_affine.py:281-300);
    * ``scale_range`` — ``(min, max)`` sampled independently per axis, or three ``(min, max)`` pairs in (Z, Y, X) order;
      ``isotropic_scale`` reuses the first axis' draw; ``safe_crop_size`` / ``safe_crop_coverage`` raise the scale to the floor
      that keeps a following centre crop inside the source (_affine.py:310-357);
    * ``translate_range`` — voxels per axis (Z, Y, X), a value ``v`` meaning ``(-v, v)``.

    The random stream is this class' own (``generator``); the matrix composition restates kornia's
    (``kornia_affine_matrix3d``).  ``params=Minv`` (B, 3, 4: output -> input voxel) injects the matrices."""

    is_spatial = True

    align_corners = False

    def __init__(self, keys, prob: float = 0.1, rotate_range=None, shear_range=None, translate_range=None, scale_range=None,
                 isotropic_scale: bool = False, scale_z_shear: bool = True, mode: str = "bilinear", padding_mode: str = "zeros",
                 safe_crop_size=None, safe_crop_coverage: float = 1.0, allow_missing_keys: bool = False):
        super().__init__(keys, prob)
        if padding_mode not in _PADDING_MODES:
            raise ValueError(f"padding_mode must be one of {sorted(_PADDING_MODES)}, got {padding_mode!r}")
        self.mode, self.padding_mode = mode, padding_mode

        def pairs(v, n):  # per-axis (Z, Y, X) values -> kornia (X, Y, Z) list of (lo, hi)
            if v is None:
                return [(0.0, 0.0)] * n
            if isinstance(v, (int, float)):
                v = [v] * n
            out = [(float(a[0]), float(a[1])) if isinstance(a, (tuple, list)) else (-float(a), float(a)) for a in v]
            return list(reversed(out))

        rr = pairs(rotate_range, 3)
        self.degrees = [(math.degrees(lo), math.degrees(hi)) for lo, hi in rr]
        self.translate = pairs(translate_range, 3)
        # shears: kornia facet order (sxy, sxz, syx, syz, szx, szy), degrees
        if shear_range is None:
            self.shears = [(0.0, 0.0)] * 6
        elif isinstance(shear_range, (int, float)):
            self.shears = [(-float(shear_range), float(shear_range))] * 6
        elif len(shear_range) == 2 and not isinstance(shear_range[0], (list, tuple)):
            self.shears = [(float(shear_range[0]), float(shear_range[1]))] * 6
        elif len(shear_range) == 6:
            self.shears = [(float(p_[0]), float(p_[1])) if isinstance(p_, (list, tuple)) else (-float(p_), float(p_)) for p_ in shear_range]
        elif len(shear_range) == 3 and not isinstance(shear_range[0], (list, tuple)):
            s_zy, s_zx, s_yz = (float(v) for v in shear_range)
            self.shears = [(0.0, 0.0)] * 3 + [(-s_yz, s_yz), (-s_zx, s_zx), (-s_zy, s_zy)]
        else:
            raise ValueError(f"shear_range must be (min, max), [s_zy, s_zx, s_yz] (3-value), or 6 (min, max) pairs. Got {shear_range!r}.")
        per_axis = False
        if scale_range is None:
            self.scale = None
        elif len(scale_range) == 3 and isinstance(scale_range[0], (list, tuple)):
            z, y, x = scale_range
            self.scale, per_axis = [tuple(map(float, x)), tuple(map(float, y)), tuple(map(float, z))], True
        elif len(scale_range) == 2:
            self.scale = [(float(scale_range[0]), float(scale_range[1]))] * 3
        else:
            raise ValueError(f"scale_range must be (min, max) or [(z_min, z_max), (y_min, y_max), (x_min, x_max)]. "
                             f"Got {scale_range!r} with length {len(scale_range)}.")
        if isotropic_scale and per_axis:
            raise ValueError("isotropic_scale=True cannot be combined with per-axis scale_range. Use a flat (min, max) range instead.")
        self.isotropic = bool(isotropic_scale and self.scale is not None)
        self.scale_z_shear = scale_z_shear
        self.safe_crop_size = tuple(safe_crop_size) if safe_crop_size is not None else None
        self.safe_crop_coverage = safe_crop_coverage

    def _uniform(self, B: int, ranges) -> Tensor:
        lo = torch.tensor([r[0] for r in ranges], dtype=torch.float64)
        hi = torch.tensor([r[1] for r in ranges], dtype=torch.float64)
        return lo + (hi - lo) * torch.rand(B, len(ranges), generator=self.generator, dtype=torch.float64)

    def sample_parameters(self, shape) -> dict:
        """kornia-style parameter block: angles / scale / shears / translations in (x, y, z) order + the apply mask"""
        B, _, D, H, W = shape
        prm = {"apply": self._rand(B) < self.prob, "angles": self._uniform(B, self.degrees),
               "translations": self._uniform(B, self.translate), "shears": self._uniform(B, self.shears),
               "scale": self._uniform(B, self.scale) if self.scale is not None else torch.ones(B, 3, dtype=torch.float64)}
        if self.isotropic:
            prm["scale"] = prm["scale"][:, :1].expand(-1, 3).clone()
        if self.safe_crop_size is not None:
            floor = self._compute_scale_floor(prm["angles"], shape, self.safe_crop_size) * self.safe_crop_coverage
            if self.isotropic:
                floor = floor.max(dim=-1, keepdim=True).values.expand_as(floor)
            prm["scale"] = torch.max(prm["scale"], floor)
        if self.scale_z_shear and max(H, W) > 1 and D < max(H, W):  # _scale_z_shear_facets: sxz, syz, szx, szy
            prm["shears"][:, [1, 3, 4, 5]] *= D / max(H, W)
        return prm

    @staticmethod
    def _compute_scale_floor(angles: Tensor, input_shape, safe_crop_size) -> Tensor:
        """_affine.py:310-357: per-axis minimum scale (kornia X, Y, Z order) such that a centre crop of ``safe_crop_size``
        (Z, Y, X) behind a Z rotation by ``angles[:, 2]`` degrees samples inside the source"""
        th = torch.deg2rad(angles[:, 2])
        c, s_ = th.cos().abs(), th.sin().abs()
        dz, dy, dx = (v / 2.0 for v in safe_crop_size)
        hz, hy, hx = input_shape[2] / 2.0, input_shape[3] / 2.0, input_shape[4] / 2.0
        return torch.stack([(c * dx + s_ * dy) / hx, (s_ * dx + c * dy) / hy, torch.full_like(c, dz / hz)], dim=-1)

    def randomize(self, shape) -> Tensor:
        """output -> input voxel matrices (B, 3, 4); identity where the transform is not applied"""
        B, _, D, H, W = shape
        prm = self.sample_parameters(shape)
        M = kornia_affine_matrix3d(prm["angles"], prm["scale"], prm["shears"], prm["translations"], (D, H, W))
        M = torch.where(prm["apply"].view(-1, 1, 1), M, torch.eye(4, dtype=torch.float64).expand(B, 4, 4))
        return torch.linalg.inv(M)[:, :3].float().contiguous()

    def __call__(self, sample: dict, params: Tensor | None = None, crop_roi_size=None) -> dict:
        """``crop_roi_size``: produce only the centre window a following ``BatchedCenterSpatialCropd(roi_size)`` would
        keep (identical voxels, ~3x fewer of them in the recipes) — see ``fuse_affine_crop``."""
        first = next((k for k in self.keys if k in sample), None)
        if first is None:
            return sample
        Minv = params if params is not None else self.randomize(sample[first].shape)
        for k in self.keys:
            if k in sample:
                win = _center_window(sample[k].shape, crop_roi_size) if crop_roi_size is not None else None
                sample[k] = warp_affine3d(sample[k], Minv, self.mode, win, self.padding_mode, self.align_corners)
        return sample


class _AffineThenCenterCrop:
    """``BatchedRandAffined`` immediately followed by ``BatchedCenterSpatialCropd`` over the same keys, as one pass."""

    def __init__(self, affine: "BatchedRandAffined", crop: "BatchedCenterSpatialCropd"):
        self.affine, self.crop = affine, crop

    def __call__(self, sample: dict) -> dict:
        return self.affine(sample, crop_roi_size=self.crop.roi_size)


def fuse_affine_crop(transforms: Sequence) -> list:
    """Peephole over a GPU augmentation chain (viscy_data/hcs.py:694-695 applies it in order): an affine whose output is
    centre-cropped next (every fit recipe: oversized patch -> affine -> crop to the training size) only computes the
    voxels that survive the crop.  Same values, same RNG consumption; anything else passes through untouched."""
    out, i = [], 0
    ts = list(transforms)
    while i < len(ts):
        t = ts[i]
        nxt = ts[i + 1] if i + 1 < len(ts) else None
        if (isinstance(t, BatchedRandAffined) and isinstance(nxt, BatchedCenterSpatialCropd) and len(nxt.roi_size) == 3
                and list(nxt.keys) == list(t.keys)):
            out.append(_AffineThenCenterCrop(t, nxt))
            i += 2
        else:
            out.append(t)
            i += 1
    return out


# ------------------------------------------------------------------------------------------------
# RandWeightedCropd — the CPU-worker multi-sample crop of the fit recipes (viscy_transforms/_monai_wrappers.py:142-183, a
# jsonargparse-friendly subclass of MONAI's RandWeightedCropd; MONAI 1.5.2 is not under /root/reference: the sampling below
# restates its published ``weighted_patch_samples`` / ``SpatialCrop`` — "parity unpinned" for the third-party internals).
# Host code by design: it runs inside DataLoader workers on per-FOV stacks before batching (hcs.py:783); the batched device
# version is BatchedRandWeightedCropd below.
# ------------------------------------------------------------------------------------------------
class RandWeightedCropd:
    """``num_samples`` crops of ``spatial_size`` per sample, centres drawn with probability proportional to the weight map
    ``sample[w_key]`` (1, Z, Y, X) restricted to the centres whose window fits; ``-1`` / non-positive entries of
    ``spatial_size`` keep the whole axis.  Returns a list of ``num_samples`` dicts (MONAI's multi-sample convention, which
    ``HCSDataModule`` flattens: train batch = ``batch_size // num_samples`` stacks)."""

    is_spatial = True

    def __init__(self, keys, w_key: str, spatial_size: Sequence[int], num_samples: int = 1, allow_missing_keys: bool = False,
                 lazy: bool = False):
        self.keys, self.w_key = _keys(keys), w_key
        self.spatial_size, self.num_samples = tuple(int(s) for s in spatial_size), int(num_samples)
        self.allow_missing_keys = allow_missing_keys
        import numpy as _np

        self.R = _np.random.RandomState()

    def set_random_state(self, seed: int | None = None, state=None):
        import numpy as _np

        self.R = state if state is not None else _np.random.RandomState(seed)
        return self

    def _centers(self, w: Tensor) -> list[tuple[int, ...]]:
        import numpy as _np

        img_size = tuple(w.shape)
        win = tuple(m if s <= 0 else min(s, m) for s, m in zip(self.spatial_size, img_size))  # fall_back_tuple
        sl = tuple(slice(k // 2, m - k + k // 2) if m > k else slice(m // 2, m // 2 + 1) for k, m in zip(win, img_size))
        v = _np.asarray(w[sl], dtype=_np.float64)
        v_size = v.shape
        v = v.ravel().copy()
        v[~_np.isfinite(v)] = 0.0
        if (v < 0).any():
            v -= v.min()  # shift to non-negative
        c = _np.cumsum(v)
        if not c[-1] or not _np.isfinite(c[-1]) or c[-1] < 0:  # uniform sampling
            idx = self.R.randint(0, len(c), size=self.num_samples)
        else:
            r = self.R.random(self.num_samples)
            idx = _np.minimum(_np.searchsorted(c, r * c[-1], side="right"), len(c) - 1)
        diff = [min(k, m) // 2 for k, m in zip(win, img_size)]
        return [tuple(int(u) + d for u, d in zip(_np.unravel_index(int(i), v_size), diff)) for i in idx], win

    def __call__(self, sample: dict) -> list[dict]:
        w = sample[self.w_key]
        centers, win = self._centers(w[0])
        out = []
        for ctr in centers:
            sl = tuple(slice(max(c - k // 2, 0), max(c - k // 2, 0) + k) for c, k in zip(ctr, win))  # SpatialCrop(roi_center, roi_size)
            d = dict(sample)
            for k in self.keys:
                if k not in sample:
                    if self.allow_missing_keys:
                        continue
                    raise KeyError(k)
                d[k] = sample[k][(slice(None),) + sl]
            out.append(d)
        return out


# ------------------------------------------------------------------------------------------------
# K23 BatchedRandWeightedCropd (_crop.py:263-386)
# ------------------------------------------------------------------------------------------------
def crop3d(x: Tensor, starts: Tensor, size: Sequence[int]) -> Tensor:
    """per-sample (Z, Y, X) crops of a (B, C, Z, Y, X) batch at ``starts`` (B, 3) — one gather launch"""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 5):
        raise RuntimeError("viscy_amd GPU augmentations need a contiguous float32 (B,C,Z,Y,X) batch on the HIP device (no CPU fallback)")
    B, C, Z, Y, X = x.shape
    cz, cy, cx = (int(v) for v in size)
    st = starts.to(x.device, torch.int32).contiguous()
    y = torch.empty((B, C, cz, cy, cx), dtype=torch.float32, device=x.device)
    check(lib().vsx_crop3d(ptr(x), ptr(y), ptr(st), B, C, Z, Y, X, cz, cy, cx, stream()), "crop3d")
    return y


class BatchedRandWeightedCropd(_BatchedRand):
    """Crop positions drawn with probability proportional to the weight map summed over each candidate window; every key
    is cropped at the same per-sample position.  ``params=(z_starts, y_starts, x_starts)`` injects the positions;
    ``uniforms=(u_yx, u_z)`` injects the random numbers of the inverse-CDF draw (torch.multinomial's stream cannot be
    reproduced)."""

    is_spatial = True

    def __init__(self, keys, w_key: str, spatial_size: Sequence[int], allow_missing_keys: bool = False):
        super().__init__(keys, 1.0)
        self.w_key, self._spatial_size = w_key, tuple(int(v) for v in spatial_size)
        self.allow_missing_keys = allow_missing_keys

    def sample_crop_starts(self, weight_map: Tensor, uniforms=None):
        if not _gpu_ok(weight_map) or weight_map.ndim != 5:
            raise RuntimeError("viscy_amd GPU augmentations need a contiguous float32 (B,C,Z,Y,X) batch on the HIP device (no CPU fallback)")
        B, C, Z, Y, X = weight_map.shape
        cz, cy, cx = self._spatial_size
        dev = weight_map.device
        vy, vx = Y - cy + 1, X - cx + 1
        wpool = torch.empty((B, vy * vx), dtype=torch.float32, device=dev)
        tmp = torch.empty(B * Y * X + B * Y * vx, dtype=torch.float32, device=dev)
        check(lib().vsx_crop_weights(ptr(weight_map), ptr(wpool), ptr(tmp), B, C * Z, Y, X, cy, cx, stream()), "crop_weights")
        u_yx, u_z = uniforms if uniforms is not None else (self._rand(B), self._rand(B))
        u_d = u_yx.to(dev, torch.float32).contiguous()
        idx = torch.empty(B, dtype=torch.int32, device=dev)
        check(lib().vsx_sample_index(ptr(wpool), ptr(u_d), ptr(idx), B, vy * vx, stream()), "sample_index")
        idx = idx.long()
        y_starts, x_starts = idx // vx, idx % vx
        if cz >= Z:
            z_starts = torch.zeros(B, dtype=torch.long, device=dev)
        else:
            z_starts = (u_z.to(dev).double() * (Z - cz + 1)).long().clamp_(0, Z - cz)
        return z_starts, y_starts, x_starts

    def __call__(self, sample: dict, params=None, uniforms=None) -> dict:
        d = dict(sample)
        wm = d[self.w_key]
        if wm.ndim != 5:
            raise ValueError(f"BatchedRandWeightedCropd requires 5D input (B, C, Z, Y, X), got {wm.ndim}D.")
        _, _, Z, Y, X = wm.shape
        cz, cy, cx = self._spatial_size
        if cz > Z:
            raise ValueError(f"spatial_size Z ({cz}) exceeds input Z ({Z}).")
        if cy > Y or cx > X:
            raise ValueError(f"spatial_size YX ({cy}, {cx}) exceeds input YX ({Y}, {X}).")
        z0, y0, x0 = params if params is not None else self.sample_crop_starts(wm, uniforms)
        starts = torch.stack([torch.as_tensor(z0), torch.as_tensor(y0), torch.as_tensor(x0)], dim=1)
        for k in self.keys:
            if k not in d:
                if self.allow_missing_keys:
                    continue
                raise KeyError(k)
            d[k] = crop3d(d[k], starts, self._spatial_size)
        return d


class CenterSpatialCropd:
    """``viscy_transforms.CenterSpatialCropd`` (_monai_wrappers.py:510-540, MONAI's transform): host-side centre crop of the
    trailing spatial dims of (C, Z, Y, X) samples in the DataLoader workers; ``roi_size`` entries <= 0 keep the axis."""

    is_spatial = True

    def __init__(self, keys, roi_size, allow_missing_keys: bool = False, lazy: bool = False):
        self.keys, self.allow_missing_keys = _keys(keys), allow_missing_keys
        self.roi_size = roi_size

    def __call__(self, sample: dict) -> dict:
        d = dict(sample)
        for k in self.keys:
            if k not in d:
                if self.allow_missing_keys:
                    continue
                raise KeyError(k)
            x = d[k]
            nsp = x.ndim - 1
            roi = [self.roi_size] * nsp if isinstance(self.roi_size, int) else list(self.roi_size)
            sl = [slice(None)]
            for dim, size in zip(x.shape[1:], roi):
                size = dim if size <= 0 else min(size, dim)
                c = dim // 2                      # MONAI SpatialCrop(roi_center = dim // 2, roi_size)
                start = max(c - size // 2, 0)
                sl.append(slice(start, start + size))
            d[k] = x[tuple(sl)]
        return d


class BatchedRandInvertIntensityd(_BatchedRand):
    """``viscy_transforms.BatchedRandInvertIntensityd`` (_invert_intensity.py:16-80): each sample of the batch is negated
    with probability ``prob`` (one draw per sample, shared by the keys) — the scale kernel with factor -2 (x * (1 - 2))."""

    is_spatial = False

    def __init__(self, keys, prob: float = 0.1, allow_missing_keys: bool = False):
        super().__init__(keys, prob)
        self.allow_missing_keys = allow_missing_keys

    def __call__(self, sample: dict, params: Tensor | None = None) -> dict:
        first = next((k for k in self.keys if k in sample), None)
        if first is None:
            return sample
        do = params if params is not None else self._rand(sample[first].shape[0]) < self.prob
        f = torch.where(do.bool().cpu(), torch.tensor(-2.0), torch.tensor(0.0))
        for k in self.keys:
            if k in sample:
                sample[k] = intensity_augment(sample[k], factor=f)
        return sample


class BatchedStackChannelsd:
    """``viscy_transforms.BatchedStackChannelsd`` (_stack_channels.py:20-82): ``{"source": ["phase"], "target": ["nuclei",
    "membrane"]}`` -> the named single-channel (B, 1, Z, Y, X) entries concatenated along the channel axis (data movement)."""

    is_spatial = False

    def __init__(self, channel_map: dict):
        self.channel_map = {k: list(v) for k, v in channel_map.items()}

    def __call__(self, sample: dict) -> dict:
        return {key: torch.cat([sample[ch] for ch in chans], dim=1) for key, chans in self.channel_map.items()}
