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


def adjust_contrast(x: Tensor, gamma: Tensor, apply: Tensor, invert_image: bool = False, retain_stats: bool = False) -> Tensor:
    """BatchedRandAdjustContrast.__call__, _adjust_contrast.py:54-86, which loops MONAI
    ``AdjustContrast(gamma, invert_image, retain_stats)`` (monai 1.5.2 transforms/intensity/array.py — not installed here,
    restated from its published source) per sample:
    [invert: x = -x] → [retain_stats: mn = x.mean(); sd = x.std()] → eps=1e-7; m=x.min(); r=x.max()-m;
    ret = ((x-m)/(r+eps))**gamma * r + m → [retain_stats: ret -= ret.mean(); ret /= ret.std() + 1e-8; ret = sd*ret + mn]
    → [invert: ret = -ret].  ``std`` is torch's unbiased one."""
    out = torch.empty_like(x)
    for i in range(x.shape[0]):
        s = x[i]
        if bool(apply[i]):
            if invert_image:
                s = -s
            if retain_stats:
                mn, sd = s.mean(), s.std()
            m = s.min()
            r = s.max() - m
            s = ((s - m) / (r + 1e-7)) ** float(gamma[i]) * r + m
            if retain_stats:
                s = s - s.mean()
                s = s / (s.std() + 1e-8)
                s = sd * s + mn
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


def gaussian_erf_kernel1d(kernel_size: int, sigma: Tensor) -> Tensor:
    """kornia 0.8.3 ``get_gaussian_erf_kernel1d`` (called at _gaussian_smooth.py:139; kornia is not installed here —
    restated from its published source): pixel-integrated Gaussian taps
    ``0.5*(erf((i+0.5)/(σ√2)) - erf((i-0.5)/(σ√2)))``, i = -r..r, normalised to sum 1; σ is (B, 1)."""
    r = kernel_size // 2
    i = torch.arange(-r, r + 1, dtype=torch.float32).view(1, -1)
    t = 0.70710678 / sigma.float().abs().clamp_min(0.0)
    t = torch.where(torch.isfinite(t), t, torch.full_like(t, float("inf")))
    g = 0.5 * (torch.erf((i + 0.5) * t) - torch.erf((i - 0.5) * t))
    g = torch.nan_to_num(g, nan=0.0)
    g = g.clamp_min(0)
    return g / g.sum(-1, keepdim=True)


def estimate_kernel_size(sigma: float, truncated: float = 4.0) -> int:
    """_gaussian_smooth.py:119-122."""
    tail = int(max(float(sigma) * truncated, 0.5) + 0.5)
    return 2 * tail + 1


def gaussian_smooth(x: Tensor, sigma: Tensor, apply: Tensor, truncated: float = 4.0) -> Tensor:
    """BatchedRandGaussianSmooth.__call__, _gaussian_smooth.py:141-167, with injected per-sample sigma (B, 3) in
    (Z, Y, X) order: separable filter3d with zero ("constant") border; kernel size from the max sigma of the
    selected samples per axis."""
    import torch.nn.functional as F

    out = x.clone()
    idx = torch.where(apply.bool())[0]
    if len(idx) == 0:
        return out
    data = x[idx].float()
    sg = sigma[idx].float()
    B, C = data.shape[:2]
    for axis in range(3):
        s = sg[:, axis]
        if not (s > 0).any():
            continue
        k = estimate_kernel_size(s.max().item(), truncated)
        taps = gaussian_erf_kernel1d(k, s.view(-1, 1))  # (B, k)
        shape = [1, 1, 1]
        shape[axis] = k
        pad = [0, 0, 0]
        pad[axis] = k // 2
        w = taps.repeat_interleave(C, 0).view(B * C, 1, *shape)
        data = F.conv3d(data.reshape(1, B * C, *data.shape[2:]), w, padding=pad, groups=B * C).view_as(data)
    out[idx] = data.to(x.dtype)
    return out


def warp_affine3d(x: Tensor, Minv: Tensor, mode: str = "bilinear", padding_mode: str = "zeros", align_corners: bool = False) -> Tensor:
    """What kornia ``warp_affine3d(..., padding_mode=..., align_corners=...)`` (called at _affine.py:33-47) computes, stated with
    torch's own ``affine_grid`` / ``grid_sample``.  ``Minv`` maps output voxels (x, y, z) to input voxels in kornia's pixel frame;
    kornia normalises it with (size - 1) denominators (``normalize_homography3d``) WHATEVER ``align_corners`` is and hands the flag
    to ``affine_grid`` and ``grid_sample``.  The reference forwards ``flags["align_corners"]`` of ``RandomAffine3D`` and never sets
    it: kornia's default there is False (third-party, absent from /root/reference: restated from its published source, unpinned —
    ADVICE r4).  With True, output voxel i samples ``Minv·(i, 1)`` exactly; with False see viscy_amd.transforms.kornia_sampling_matrix."""
    import torch.nn.functional as F

    B, C, D, H, W = x.shape

    def norm_mat(d, h, w):  # voxel → [-1, 1], the (size - 1) convention kornia uses for the matrix
        return torch.tensor([[2.0 / max(w - 1, 1), 0, 0, -1.0], [0, 2.0 / max(h - 1, 1), 0, -1.0],
                             [0, 0, 2.0 / max(d - 1, 1), -1.0], [0, 0, 0, 1.0]])

    N = norm_mat(D, H, W)
    M4 = torch.cat([Minv.float(), torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(B, 1, 4)], dim=1)
    theta = (N @ M4 @ torch.linalg.inv(N))[:, :3]
    grid = F.affine_grid(theta, (B, C, D, H, W), align_corners=align_corners)
    return F.grid_sample(x.float(), grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners)


def affine_matrix_zyx(angle_z_deg: Tensor, scale_xyz: Tensor, shape_dhw, shear_xy: Tensor | None = None) -> Tensor:
    """Output→input voxel matrix for a rotation about the Z axis (in the YX plane) by ``angle`` degrees, per-axis
    scale and optional XY shear about the volume centre — the parameterisation the UNeXt2 recipes use
    (rotate_range only about Z, _affine.py:248-276; applications/dynacell/.../unext2_fit.yml)."""
    D, H, W = shape_dhw
    B = angle_z_deg.shape[0]
    th = torch.deg2rad(angle_z_deg.float())
    c, s = th.cos(), th.sin()
    A = torch.zeros(B, 4, 4)
    A[:, 0, 0], A[:, 0, 1], A[:, 1, 0], A[:, 1, 1], A[:, 2, 2], A[:, 3, 3] = c, -s, s, c, 1.0, 1.0
    S = torch.diag_embed(torch.cat([scale_xyz.float(), torch.ones(B, 1)], dim=1))
    Sh = torch.eye(4).repeat(B, 1, 1)
    if shear_xy is not None:
        Sh[:, 0, 1] = shear_xy.float()
    ctr = torch.eye(4).repeat(B, 1, 1)
    ctr[:, 0, 3], ctr[:, 1, 3], ctr[:, 2, 3] = (W - 1) / 2.0, (H - 1) / 2.0, (D - 1) / 2.0
    fwd = ctr @ A @ Sh @ S @ torch.linalg.inv(ctr)  # input → output
    return torch.linalg.inv(fwd)[:, :3]


def weighted_crop_window_weights(weight_map: Tensor, crop_yx) -> Tensor:
    """BatchedRandWeightedCropd._sample_crop_starts, _crop.py:317-333: (B, vy*vx) pooled window weights, all-zero maps -> 1."""
    import torch.nn.functional as F

    w = weight_map.sum(dim=(1, 2)).clamp(min=0).float()
    wp = F.avg_pool2d(w.unsqueeze(1), tuple(crop_yx), stride=1)
    flat = wp.view(wp.shape[0], -1).clone()
    flat[flat.sum(dim=1) == 0] = 1.0
    return flat


def inverse_cdf_index(weights: Tensor, u: Tensor) -> Tensor:
    """the draw torch.multinomial(weights, 1) makes, written with explicit uniforms: smallest i with cdf_i > u * total"""
    cdf = weights.double().cumsum(1)
    target = u.double().view(-1, 1) * cdf[:, -1:]
    return torch.searchsorted(cdf, target, right=True).squeeze(1).clamp_max(weights.shape[1] - 1)


def crop3d(img: Tensor, z0, y0, x0, size) -> Tensor:
    """_crop.py:368-384"""
    cz, cy, cx = size
    return torch.stack([img[b, :, z0[b] : z0[b] + cz, y0[b] : y0[b] + cy, x0[b] : x0[b] + cx] for b in range(img.shape[0])])


def kornia_affine_matrix3d(angles_xyz_deg: Tensor, scale_xyz: Tensor, shears_deg: Tensor, translations_xyz: Tensor, shape_dhw) -> Tensor:
    """What kornia 0.8.3 ``RandomAffine3D`` applies (``get_affine_matrix3d``; kornia is neither under /root/reference nor
    installed: restated from its published source, "parity unpinned"), as a product of elementary homogeneous matrices:
    T(c) · R(-angles as rotation vector) · diag(scale) · T(-c), translation added, then the shear matrix about c.
    Argument conventions as BatchedRandAffined builds them (_affine.py:165-276): kornia (x, y, z) order, degrees."""
    D, H, W = shape_dhw
    B = angles_xyz_deg.shape[0]
    c = torch.tensor([(W - 1) / 2.0, (H - 1) / 2.0, (D - 1) / 2.0], dtype=torch.float64)
    out = []
    for b in range(B):
        w = torch.deg2rad(-angles_xyz_deg[b].double())
        K = torch.tensor([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]], dtype=torch.float64)
        R = torch.linalg.matrix_exp(K)  # rotation vector -> matrix
        A = torch.eye(4, dtype=torch.float64)
        A[:3, :3] = R @ torch.diag(scale_xyz[b].double())
        Tc, Tn = torch.eye(4, dtype=torch.float64), torch.eye(4, dtype=torch.float64)
        Tc[:3, 3], Tn[:3, 3] = c, -c
        M = Tc @ A @ Tn
        M[:3, 3] += translations_xyz[b].double()
        sxy, sxz, syx, syz, szx, szy = torch.tan(torch.deg2rad(shears_deg[b].double()))
        m11 = sxy * syx + 1
        m21 = sxz * syx + syz
        S3 = torch.tensor([[1.0, -syx, -(syx * szy + szx)],
                           [-sxy, m11, -(sxy * szx + szy * m11)],
                           [-sxz, -m21, sxz * szx + szy * m21 + 1]], dtype=torch.float64)
        S = torch.eye(4, dtype=torch.float64)
        S[:3, :3] = S3
        # translation column that kornia derives from the centre
        S[0, 3] = syx * c[1] + (syx * szy + szx) * c[2]
        S[1, 3] = sxy * c[0] + m11 * c[1] + (sxy * szx + szy * m11) * c[2] - c[1]
        S[2, 3] = sxz * c[0] + m21 * c[1] + (sxz * szx + szy * m21 + 1) * c[2] - c[2]
        out.append(M @ S)
    return torch.stack(out)
