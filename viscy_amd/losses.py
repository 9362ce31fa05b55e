"""MixedLoss on MI355X — drop-in for ``viscy_utils.losses.MixedLoss``
(/root/reference/packages/viscy-utils/src/viscy_utils/losses/mixed_loss.py:13-69) with the
MS-SSIM-2.5D of ``viscy_utils.evaluation.metrics.ms_ssim_25d`` (metrics.py:308-349, clamp=True).

Same constructor (l1_alpha, l2_alpha, ms_dssim_alpha), same ``forward(preds, target) -> scalar``;
the whole loss (L1 / L2 sums, 5-scale separable box-filter SSIM, pooling, data-range max) and its
gradient run in the HIP kernels of viscy_amd/csrc/loss.hip.  No CPU / eager fallback.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from . import _lib as L
from ._lib import check, lib, ptr, stream

BETAS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


_NPIX: dict = {}


def _npix_cached(dims, C, dev):
    """per-scale pixel counts of the SSIM maps, kept on the device (no H2D copy inside a captured step)"""
    key = (dims, C, str(dev))
    t = _NPIX.get(key)
    if t is None:
        t = torch.tensor([float(C * (h - 10) * (w - 10)) for (h, w) in dims] + [1.0] * (5 - len(dims)), dtype=torch.float32).to(dev)
        _NPIX[key] = t
    return t


class _MixedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds: Tensor, target: Tensor, a1: float, a2: float, a3: float):
        if preds.shape != target.shape or preds.ndim != 5:
            raise ValueError(f"preds/target must be (B, C, D, H, W) of equal shape, got {preds.shape} / {target.shape}")
        P0 = preds.detach().float().contiguous()
        T0 = target.detach().float().contiguous()
        B, C, D, H, W = P0.shape
        dev = P0.device
        ns = len(BETAS) if a3 else 0
        if a3 and (H // 16 < 11 or W // 16 < 11):
            raise ValueError(f"MS-SSIM with 5 scales needs Y, X >= 176 (got {H}x{W}): the 11x11 window must fit at 1/16 scale")
        l = lib()
        s = stream()
        need_grad = bool(getattr(ctx, "needs_input_grad", (True,))[0])
        fused = bool(ns and need_grad and l.vsx_get_flag(b"loss_fused"))
        # scalar accumulators, zero / -inf by vsx_fill_f32 (the captured step holds no ATen fill):
        #   [l1sum | l2sum | tmax x 5 (-inf), pad | sum_ssim 5B | sum_cs 5B]; a "scalar" is one float on the two-pass path and
        #   LOSS_SLOTS partial values LOSS_SLOT_STRIDE floats apart on the one-pass path (vsx_ssim_scale_fwd_fused)
        w1 = L.LOSS_SLOTS * L.LOSS_SLOT_STRIDE if fused else 4
        scal = torch.empty(8 * w1 + 10 * B, dtype=torch.float32, device=dev)
        check(l.vsx_fill_f32(ptr(scal), scal.numel(), 0.0, s), "fill")
        l1sum, l2sum = scal[0:w1], scal[w1 : 2 * w1]
        tmax = [scal[(2 + i) * w1 : (3 + i) * w1] for i in range(5)]
        check(l.vsx_fill_f32(ptr(scal[2 * w1 : 8 * w1]), 6 * w1, float("-inf"), s), "fill")
        sum_ssim = scal[8 * w1 : 8 * w1 + 5 * B]
        sum_cs = scal[8 * w1 + 5 * B : 8 * w1 + 10 * B]
        Ps, Ts, dims = [P0], [T0], [(H, W)]
        planes = B * C * D
        nlev = max(ns, 1)
        for sc in range(nlev - 1):
            h, w = dims[sc]
            Ps.append(torch.empty((B, C, D, h // 2, w // 2), dtype=torch.float32, device=dev))
            Ts.append(torch.empty_like(Ps[-1]))
            dims.append((h // 2, w // 2))
        npix_d = _npix_cached(tuple(dims), C, dev)
        dmus = []
        if fused:
            # a forward that will be differentiated: ONE pass per scale (vsx_ssim_scale_fwd_fused) gives the scale's sums, its
            # unscaled gradient field (the backward is then one pass per scale instead of two), the pooled stacks and data range
            # of the next scale and, at full resolution, the L1 / L2 sums; only the first data range needs a pass of its own
            check(l.vsx_loss_tmax(ptr(T0), T0.numel(), ptr(tmax[0]), s), "loss_tmax")
            for sc in range(ns):
                h, w = dims[sc]
                last = sc == ns - 1
                dmu = torch.empty(3 * B * C * (h - 10) * (w - 10), dtype=torch.float32, device=dev)
                check(l.vsx_ssim_scale_fwd_fused(ptr(Ps[sc]), ptr(Ts[sc]), ptr(tmax[sc]),
                                                 ptr(sum_ssim[sc * B : (sc + 1) * B]), ptr(sum_cs[sc * B : (sc + 1) * B]), ptr(dmu),
                                                 None if last else ptr(Ps[sc + 1]), None if last else ptr(Ts[sc + 1]),
                                                 None if last else ptr(tmax[sc + 1]),
                                                 ptr(l1sum) if sc == 0 else None, ptr(l2sum) if sc == 0 else None,
                                                 B, C, D, h, w, 1 if last else 0, s), "ssim_scale_fwd_fused")
                dmus.append(dmu)
        else:
            for sc in range(nlev):
                h, w = dims[sc]
                last = sc == nlev - 1
                check(l.vsx_loss_pool(ptr(Ps[sc]), ptr(Ts[sc]), None if last else ptr(Ps[sc + 1]), None if last else ptr(Ts[sc + 1]),
                                      ptr(tmax[sc]), ptr(l1sum) if sc == 0 else None, ptr(l2sum) if sc == 0 else None,
                                      planes, h, w, s), "loss_pool")
            for sc in range(ns):
                h, w = dims[sc]
                if need_grad:  # loss_fused = 0: the two-pass forward (pooling pass above, SSIM sums + gradient field here)
                    dmu = torch.empty(3 * B * C * (h - 10) * (w - 10), dtype=torch.float32, device=dev)
                    check(l.vsx_ssim_scale_fwd_dmu(ptr(Ps[sc]), ptr(Ts[sc]), ptr(tmax[sc]),
                                                   ptr(sum_ssim[sc * B : (sc + 1) * B]), ptr(sum_cs[sc * B : (sc + 1) * B]), ptr(dmu),
                                                   B, C, D, h, w, 1 if sc == ns - 1 else 0, s), "ssim_scale_fwd_dmu")
                    dmus.append(dmu)
                else:
                    check(l.vsx_ssim_scale_fwd(ptr(Ps[sc]), ptr(Ts[sc]), ptr(tmax[sc]), ptr(sum_ssim[sc * B : (sc + 1) * B]),
                                               ptr(sum_cs[sc * B : (sc + 1) * B]), B, C, D, h, w, s), "ssim_scale_fwd")
        out = torch.empty(2, dtype=torch.float32, device=dev)
        coef = torch.empty(max(ns, 1) * B * 2, dtype=torch.float32, device=dev)
        nelem = float(P0.numel())
        nslot = L.LOSS_SLOTS if fused else 1
        check(l.vsx_loss_finalize(ptr(sum_ssim), ptr(sum_cs), ptr(l1sum), ptr(l2sum), ptr(npix_d), nelem, B, max(ns, 1), nslot,
                                  a1, a2, a3, None, ptr(out[0:1]), ptr(coef), ptr(out[1:2]), s), "loss_finalize")
        ctx.saved = (Ps, Ts, dims, (w1, nslot), scal, npix_d, (B, C, D), (a1, a2, a3), ns, nelem, preds.dtype)
        # The unscaled gradient fields live from the forward to the backward: 3 * B * C * (h - 10) * (w - 10) fp32 per scale,
        # 4/3 of the full-resolution one in total = 0.99 GB at B = 512, C = 2, 256 x 256 (0.6 % of the step's 164 GB peak) — the
        # price of one pass over the stacks less per scale.  They are scratch this Function alone writes and reads (detached,
        # never exposed): plain attributes, not autograd-saved tensors, so that no version counter is involved.
        ctx.dmus = dmus
        ctx.ms_ssim = out[1]
        return out[0]

    @staticmethod
    def backward(ctx, gout: Tensor):
        Ps, Ts, dims, (w1, nslot), scal, npix_d, (B, C, D), (a1, a2, a3), ns, nelem, in_dtype = ctx.saved
        dev = Ps[0].device
        l, s = lib(), stream()
        go = gout.detach().float().reshape(1).contiguous()  # stays on the device: no host sync
        sum_ssim = scal[8 * w1 : 8 * w1 + 5 * B]
        sum_cs = scal[8 * w1 + 5 * B : 8 * w1 + 10 * B]
        coef = torch.empty(max(ns, 1) * B * 2, dtype=torch.float32, device=dev)
        tmp = torch.empty(2, dtype=torch.float32, device=dev)
        check(l.vsx_loss_finalize(ptr(sum_ssim), ptr(sum_cs), ptr(scal[0:w1]), ptr(scal[w1 : 2 * w1]), ptr(npix_d), nelem, B,
                                  max(ns, 1), nslot, a1, a2, a3, ptr(go), ptr(tmp[0:1]), ptr(coef), ptr(tmp[1:2]), s), "loss_finalize")
        dmus = getattr(ctx, "dmus", None) or []
        dnext = None
        for sc in range(max(ns, 1) - 1, -1, -1):
            h, w = dims[sc]
            dP = torch.empty((B, C, D, h, w), dtype=torch.float32, device=dev)
            l1c = a1 / nelem if sc == 0 else 0.0
            l2c = a2 / nelem if sc == 0 else 0.0
            check(l.vsx_ssim_scale_bwd_in(ptr(Ps[sc]), ptr(Ts[sc]), ptr(dmus[sc]) if ns else None,
                                          ptr(coef[sc * B * 2 : (sc + 1) * B * 2]) if ns else None, ptr(dnext), ptr(dP),
                                          B, C, D, h, w, l1c, l2c, ptr(go), 1 if ns else 0, 1 if sc == ns - 1 else 0, s),
                  "ssim_scale_bwd_in")
            dnext = dP
        return dnext.to(in_dtype), None, None, None, None


class MixedLoss(nn.Module):
    """Mixed reconstruction loss (Zhao et al.), see module docstring."""

    def __init__(self, l1_alpha: float = 0.5, l2_alpha: float = 0.0, ms_dssim_alpha: float = 0.5):
        super().__init__()
        if not any([l1_alpha, l2_alpha, ms_dssim_alpha]):
            raise ValueError("Loss term weights cannot be all zero!")
        self.l1_alpha = l1_alpha
        self.l2_alpha = l2_alpha
        self.ms_dssim_alpha = ms_dssim_alpha

    def forward(self, preds: Tensor, target: Tensor) -> Tensor:
        if not preds.is_cuda:
            raise RuntimeError("viscy_amd.MixedLoss runs on MI355X HIP kernels only (no CPU / eager fallback)")
        L.lib()
        return _MixedLossFn.apply(preds, target, float(self.l1_alpha), float(self.l2_alpha), float(self.ms_dssim_alpha))


class _MaskedMSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds: Tensor, original: Tensor, mask: Tensor):
        from . import ops as O

        p = preds.contiguous().float()
        o = original.contiguous().float()
        m = mask.contiguous().to(torch.uint8)
        loss, acc = O.masked_mse_fwd(p, o, m)
        ctx.save_for_backward(p, o, m, acc)
        ctx.in_dtype = preds.dtype
        return loss

    @staticmethod
    def backward(ctx, gout: Tensor):
        from . import ops as O

        p, o, m, acc = ctx.saved_tensors
        return O.masked_mse_bwd(p, o, m, acc, gout.contiguous().float()).to(ctx.in_dtype), None, None


class MaskedMSELoss(nn.Module):
    """``cytoland.engine.MaskedMSELoss`` (/root/reference/applications/cytoland/src/cytoland/engine.py:104-125), the FCMAE
    pre-training loss: ``(mse(preds, original).mean(2) * mask).sum() / mask.sum()`` with ``preds`` / ``original``
    (B,C,Z,Y,X) and ``mask`` (B,1,Y,X), True where the input was hidden from the encoder.  One reduction kernel forward,
    one elementwise kernel backward (``vsx_masked_mse_*``)."""

    def forward(self, preds: Tensor, original: Tensor, mask: Tensor) -> Tensor:
        if not preds.is_cuda:
            raise RuntimeError("viscy_amd.MaskedMSELoss runs on MI355X HIP kernels only (no CPU / eager fallback)")
        if preds.shape != original.shape or preds.ndim != 5:
            raise ValueError(f"preds {tuple(preds.shape)} and original {tuple(original.shape)} must be equal (B,C,Z,Y,X) shapes")
        if tuple(mask.shape) != (preds.shape[0], 1, preds.shape[3], preds.shape[4]):
            raise ValueError(f"mask must be (B,1,Y,X) = {(preds.shape[0], 1, preds.shape[3], preds.shape[4])}, got {tuple(mask.shape)}")
        L.lib()
        return _MaskedMSEFn.apply(preds, original, mask)
