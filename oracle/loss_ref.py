"""ORACLE — test infrastructure, not product code.

CPU restatement of the reference MixedLoss = a1*L1 + a2*MSE + a3*(1 - MS-SSIM-2.5D)
(/root/reference/packages/viscy-utils/src/viscy_utils/losses/mixed_loss.py:13-69 and
 .../evaluation/metrics.py:174-349).  Pinned against the reference functions themselves
(stub-imported) by oracle/validate_against_reference.py → tests/golden/loss_*.pt.

Rounding points that matter for parity (metrics.py:243-255): the five window means are
computed by a *bf16* depthwise conv3d with a bf16 uniform kernel; squared / cross
products are formed in fp32 and then rounded to bf16; conv outputs are bf16; everything
after the conv is fp32.
"""

from __future__ import annotations

from math import prod
from typing import Sequence

import torch
import torch.nn.functional as F
from torch import Tensor


def compute_ssim_and_cs_bf16(y_pred: Tensor, y: Tensor, kernel_size: Sequence[int], data_range, k1=0.01, k2=0.03):
    """metrics.py:174-269."""
    c = y_pred.size(1)
    kernel = (torch.ones((c, 1, *kernel_size), dtype=torch.float32) / float(prod(kernel_size))).to(torch.bfloat16)
    xp, yp = y_pred.float(), y.float()
    terms = [y_pred.to(torch.bfloat16), y.to(torch.bfloat16), (xp * xp).to(torch.bfloat16),
             (yp * yp).to(torch.bfloat16), (xp * yp).to(torch.bfloat16)]
    mu_x, mu_y, mu_xx, mu_yy, mu_xy = [F.conv3d(t, kernel, groups=c).float() for t in terms]
    c1 = (k1 * data_range) ** 2
    c2 = (k2 * data_range) ** 2
    sigma_x = mu_xx - mu_x * mu_x
    sigma_y = mu_yy - mu_y * mu_y
    sigma_xy = mu_xy - mu_x * mu_y
    cs = (2 * sigma_xy + c2) / (sigma_x + sigma_y + c2)
    ssim = ((2 * mu_x * mu_y + c1) / (mu_x * mu_x + mu_y * mu_y + c1)) * cs
    return ssim, cs


def ssim_25d(preds: Tensor, target: Tensor, window=(11, 11)):
    """metrics.py:272-305 with return_contrast_sensitivity=True."""
    depth = preds.shape[2]
    ssim_img, cs_img = compute_ssim_and_cs_bf16(preds, target, (depth, *window), data_range=target.max())
    return ssim_img.view(ssim_img.shape[0], -1).mean(1), cs_img.view(cs_img.shape[0], -1).mean(1)


def ms_ssim_25d(preds: Tensor, target: Tensor, window=(11, 11), clamp: bool = False,
                betas: Sequence[float] = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)) -> Tensor:
    """metrics.py:308-349."""
    base_min = 1e-4
    mcs = []
    ssim = None
    for _ in range(len(betas)):
        ssim, cs = ssim_25d(preds, target, window)
        if clamp:
            cs = cs.clamp(min=base_min)
        mcs.append(cs)
        preds = F.avg_pool3d(preds, (1, 2, 2))
        target = F.avg_pool3d(target, (1, 2, 2))
    if clamp:
        ssim = ssim.clamp(min=base_min)
    mcs[-1] = ssim
    stack = torch.stack(mcs)
    b = torch.tensor(betas).view(-1, 1)
    return torch.prod(stack**b, dim=0).mean()


def mixed_loss(preds: Tensor, target: Tensor, l1_alpha=0.5, l2_alpha=0.0, ms_dssim_alpha=0.5) -> Tensor:
    """mixed_loss.py:42-69."""
    if not any([l1_alpha, l2_alpha, ms_dssim_alpha]):
        raise ValueError("Loss term weights cannot be all zero!")
    loss = 0
    if l1_alpha:
        loss = loss + F.l1_loss(preds, target) * l1_alpha
    if l2_alpha:
        loss = loss + F.mse_loss(preds, target) * l2_alpha
    if ms_dssim_alpha:
        loss = loss + (1 - ms_ssim_25d(preds, target, clamp=True)) * ms_dssim_alpha
    return loss
