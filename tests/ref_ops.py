"""TEST INFRASTRUCTURE — plain-PyTorch statement of every op in ``viscy_amd.ops`` (same call
surface), written with high-level tensor ops that are independent of the HIP kernels' indexing.

Used two ways, only from tests/:
  * injected into ``viscy_amd.engine_unext2.Engine`` on CPU to prove that the hand-written
    forward/backward *schedule* (weight folding, GRN / IN backward algebra, layouts) equals
    autograd of the oracle model;
  * as the per-op reference the HIP kernels are compared with on the GPU.
"""

from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F
from torch import Tensor

A_ROWS, A_PATCH2, A_CONV3 = 0, 1, 2
PRO_NONE, PRO_GRN = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU_SQ, EPI_BIAS_RES, EPI_DZ, EPI_BIAS_STATS = 0, 1, 2, 3, 4, 5


def _gelu(x):
    return F.gelu(x.float())


def _gather(A: Tensor, M, K, lda, a_mode, gh, gw, cs, coff, pro, grn_s, grn_b, hw) -> Tensor:
    A = A.reshape(-1, lda).float()
    if a_mode == A_ROWS:
        a = A[:M, coff : coff + K]
    elif a_mode == A_PATCH2:
        B = M // (gh * gw)
        S = A.view(B, gh, 2, gw, 2, lda)[..., coff : coff + cs]
        a = S.permute(0, 1, 3, 2, 4, 5).reshape(M, 4 * cs)
    else:
        B = M // (gh * gw)
        S = A.view(B, gh, gw, lda)[..., coff : coff + cs]
        Sp = F.pad(S, (0, 0, 1, 1, 1, 1))
        taps = [Sp[:, ky : ky + gh, kx : kx + gw, :] for ky in range(3) for kx in range(3)]
        a = torch.stack(taps, dim=3).reshape(M, 9 * cs)
    if pro == PRO_GRN:
        b = torch.arange(M, device=A.device) // hw
        a = a * grn_s[b] + grn_b[None, :]
    return a


def gemm(kind, A, B, Cout, M, N, K, lda, ldb, ldc, *, dtype, a_mode=A_ROWS, gh=0, gw=0, cs=0, nz=1, a_coff=None,
         b_off=None, c_coff=None, c_mode=A_ROWS, c_cs=0, pro=PRO_NONE, grn_s=None, grn_b=None, hw=0, epi=EPI_NONE,
         bias=None, res=None, ldr=0, aux=None, ldx=0, red0=None, red1=None, colsum=None, C2=None, b_bstride=0, rscale=None):
    a_coff = list(a_coff) if a_coff else [0] * nz
    b_off = list(b_off) if b_off else [0] * nz
    c_coff = list(c_coff) if c_coff else [0] * nz
    rd = lambda t: t.to(dtype).float()  # noqa: E731  storage rounding
    for z in range(nz):
        a = rd(_gather(A, M, K, lda, a_mode, gh, gw, cs, a_coff[z], pro, grn_s, grn_b, hw))
        if kind == "tn":
            X = B.reshape(-1, ldb).float()[:M, b_off[z] : b_off[z] + N]
            Wv = Cout.reshape(-1)[c_coff[z] :].as_strided((N, K), (ldc, 1))
            Wv += X.t() @ a
            if colsum is not None:
                colsum += X.sum(0)
            continue
        bidx = torch.arange(M, device=a.device) // (hw if hw > 0 else M)
        if b_bstride:  # one weight matrix per batch sample
            acc = torch.empty(M, N)
            for bb in range(int(bidx.max().item()) + 1):
                Bw = B.reshape(-1)[b_off[z] + bb * b_bstride : b_off[z] + bb * b_bstride + N * ldb].view(N, ldb)[:, :K].float()
                acc[bidx == bb] = a[bidx == bb] @ Bw.t()
        else:
            Bw = B.reshape(-1)[b_off[z] : b_off[z] + N * ldb].view(N, ldb)[:, :K].float()
            acc = a @ Bw.t()
        nb = int(bidx.max().item()) + 1
        if epi in (EPI_BIAS, EPI_BIAS_GELU_SQ, EPI_BIAS_RES, EPI_BIAS_STATS) and bias is not None:
            acc = acc + bias[None, :]
        if epi == EPI_BIAS_RES:
            if rscale is not None:
                acc = acc * rscale[bidx][:, None]
            acc = acc + res.reshape(-1, ldr).float()[:M, :N]
        out = rd(acc)
        if epi == EPI_BIAS_GELU_SQ:
            gq = rd(_gelu(out))
            red0.index_add_(0, bidx, gq**2)
            C2.reshape(-1, ldc)[:M, c_coff[z] : c_coff[z] + N] = gq.to(C2.dtype)
        elif epi == EPI_DZ:
            gact = aux.reshape(-1, ldx).float()[:M, :N]
            red0.index_add_(0, bidx, out * gact)
            red1.index_add_(0, bidx, out)
        elif epi == EPI_BIAS_STATS:
            red0.index_add_(0, bidx, out)
            red1.index_add_(0, bidx, out * out)
        del nb
        if c_mode == A_PATCH2:
            Bn = M // (gh * gw)
            Cv = Cout.view(Bn, gh, 2, gw, 2, ldc)
            o = out.view(Bn, gh, gw, 2, 2, c_cs).permute(0, 1, 3, 2, 4, 5)
            Cv[..., c_coff[z] : c_coff[z] + c_cs] = o.to(Cout.dtype)
        else:
            if Cout is not None:  # EPI_BIAS_GELU_SQ in inference: only C2 (the activation) is kept
                Cout.reshape(-1, ldc)[:M, c_coff[z] : c_coff[z] + N] = out.to(Cout.dtype)


def gemm_z(kind, *args, nz, a_coff, b_off, c_coff, **kw):
    gemm(kind, *args, nz=nz, a_coff=a_coff, b_off=b_off, c_coff=c_coff, **kw)


def ln_fwd(x, gamma, beta, rows, Cc, eps=1e-6, need_mean=True):
    xf = x.float().view(rows, Cc)
    mean = xf.mean(1)
    var = xf.var(1, unbiased=False)
    rstd = torch.rsqrt(var + eps)
    y = (xf - mean[:, None]) * rstd[:, None]
    if gamma is not None:
        y = y * gamma + beta
    return y.to(x.dtype), (mean if need_mean else None), rstd


def ln_bwd(dy, x, mean, rstd, gamma, add, dgamma, dbeta, rows, Cc):
    dyf, xf = dy.float().view(rows, Cc), x.float().view(rows, Cc)
    xh = (xf - mean[:, None]) * rstd[:, None] if mean is not None else xf
    gch = dyf * gamma if gamma is not None else dyf
    dx = rstd[:, None] * (gch - gch.mean(1, keepdim=True) - xh * (gch * xh).mean(1, keepdim=True))
    if add is not None:
        dx = dx + add.float().view(rows, Cc)
    if dgamma is not None:
        dgamma += (dyf * xh).sum(0)
        dbeta += dyf.sum(0)
    return dx.to(dy.dtype)


def grn_scale(colsq, gamma, eps=1e-6):
    gq = colsq.sqrt()
    return 1.0 + gamma * gq / (gq.mean(1, keepdim=True) + eps)


@torch.enable_grad()
def grn_bwd_stats(colsq, P, gamma, dgamma, eps=1e-6, Sb=None, dbeta=None):
    if Sb is not None:
        dbeta += Sb.sum(0)
    # autograd through n = g / (mean g + eps) with upstream dn = gamma * P
    gq = colsq.sqrt().clone().requires_grad_(True)
    n = gq / (gq.mean(1, keepdim=True) + eps)
    (dg,) = torch.autograd.grad(n, gq, gamma * P)
    dgamma += (n.detach() * P).sum(0)
    gq = gq.detach()
    return torch.where(gq > 0, dg / gq, torch.zeros_like(dg))


@torch.enable_grad()
def grn_gelu_bwd(dz, h, s, t, colsum, M, N, hw):
    b = torch.arange(M, device=dz.device) // hw
    hf = h.float().requires_grad_(True)
    gact = F.gelu(hf)
    dG = dz.float() * s[b] + gact.detach() * t[b]
    (dh,) = torch.autograd.grad(gact, hf, dG)
    dh = dh.to(dz.dtype)
    colsum += dh.float().sum(0)
    dz.copy_(dh)


def _nhwc_to_nchw(x, B, H, W, C):
    return x.float().view(B, H, W, C).permute(0, 3, 1, 2)


def dwconv7_fwd(x, w, bias, B, H, W, Cc):
    wt = w.t().reshape(Cc, 1, 7, 7)
    y = F.conv2d(_nhwc_to_nchw(x, B, H, W, Cc), wt, bias, padding=3, groups=Cc)
    return y.permute(0, 2, 3, 1).reshape(B * H * W, Cc).to(x.dtype)


@torch.enable_grad()
def dwconv7_bwd_data(dy, w, add, B, H, W, Cc):
    wt = w.t().reshape(Cc, 1, 7, 7)
    xin = torch.zeros(B, Cc, H, W, requires_grad=True)
    y = F.conv2d(xin, wt, None, padding=3, groups=Cc)
    (dx,) = torch.autograd.grad(y, xin, _nhwc_to_nchw(dy, B, H, W, Cc))
    dx = dx.permute(0, 2, 3, 1).reshape(B * H * W, Cc)
    if add is not None:
        dx = dx + add.float().view(B * H * W, Cc)
    return dx.to(dy.dtype)


@torch.enable_grad()
def dwconv7_bwd_weight(dy, x, dw, db, B, H, W, Cc):
    wt = torch.zeros(Cc, 1, 7, 7, requires_grad=True)
    y = F.conv2d(_nhwc_to_nchw(x, B, H, W, Cc), wt, None, padding=3, groups=Cc)
    (g,) = torch.autograd.grad(y, wt, _nhwc_to_nchw(dy, B, H, W, Cc))
    dw += g.reshape(Cc, 49).t()
    if db is not None:
        db += dy.float().view(-1, Cc).sum(0)


def im2col3x3(x, B, H, W, Cc):
    img = F.pad(x.float().view(B, H, W, Cc), (0, 0, 1, 1, 1, 1))
    cols = [img[:, ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)]
    return torch.stack(cols, dim=3).reshape(B * H * W, 9 * Cc).to(x.dtype)


def col2im3x3(dcol, B, H, W, Cc):
    d = dcol.float().view(B, H, W, 9, Cc)
    out = torch.zeros(B, H + 2, W + 2, Cc)
    for t in range(9):
        ky, kx = divmod(t, 3)
        out[:, ky:ky + H, kx:kx + W, :] += d[:, :, :, t, :]
    return out[:, 1:H + 1, 1:W + 1, :].reshape(B * H * W, Cc).to(dcol.dtype)


def pad_cols(src, Kp):
    R, K = src.shape
    dst = torch.zeros((R, Kp), dtype=src.dtype)
    dst[:, :K] = src
    return dst


def stem_im2col(x, kernel, dtype, sub=None, div=None, ld=None):
    if ld is not None:
        return pad_cols(stem_im2col(x, kernel, dtype, sub, div), ld)
    B, Cin, Z, H, W = x.shape
    kz, ky, kx = kernel
    if sub is not None:
        x = (x - sub.view(B, 1, 1, 1, 1)) / (div.view(B, 1, 1, 1, 1) + 1e-8)
    D, h, w = Z // kz, H // ky, W // kx
    p = x.view(B, Cin, D, kz, h, ky, w, kx).permute(0, 4, 6, 2, 1, 3, 5, 7)  # b,h,w,d,ci,kz,ky,kx
    return p.reshape(B * h * w, D * Cin * kz * ky * kx).to(dtype)


def pixel_shuffle_cat_fwd(low, skip, B, h, w, c, cs):
    up = F.pixel_shuffle(_nhwc_to_nchw(low, B, h, w, 4 * c), 2).permute(0, 2, 3, 1)
    parts = [up]
    if cs:
        parts.append(skip.float().view(B, 2 * h, 2 * w, cs))
    return torch.cat(parts, dim=-1).reshape(B * 4 * h * w, c + cs).to(low.dtype)


def pixel_shuffle_cat_bwd(dcat, B, h, w, c, cs):
    d = dcat.float().view(B, 2 * h, 2 * w, c + cs)
    dlow = F.pixel_unshuffle(d[..., :c].permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(B * h * w, 4 * c)
    dskip = d[..., c:].reshape(B * 4 * h * w, cs).to(dcat.dtype) if cs else None
    return dlow.to(dcat.dtype), dskip


def _head_shuffle(dec_nchw, C3, D, pool):
    x = F.pixel_shuffle(dec_nchw, 2)
    if pool:
        x = F.avg_pool2d(F.pad(x, (1, 0, 1, 0)), kernel_size=2, stride=1)
    B, Cm, H2, W2 = x.shape
    # reference channel = c3*D + z  →  ours z*C3 + c3
    return x.view(B, C3, D, H2, W2).permute(0, 3, 4, 2, 1).reshape(B * H2 * W2, D * C3)


def head_shuffle_fwd(dec, B, h, w, C3, D, pool):
    return _head_shuffle(_nhwc_to_nchw(dec, B, h, w, 4 * C3 * D), C3, D, pool).to(dec.dtype)


@torch.enable_grad()
def head_shuffle_bwd(dhin, B, h, w, C3, D, pool):
    xin = torch.zeros(B, 4 * C3 * D, h, w, requires_grad=True)
    y = _head_shuffle(xin, C3, D, pool)
    (g,) = torch.autograd.grad(y, xin, dhin.float())
    return g.permute(0, 2, 3, 1).reshape(B * h * w, 4 * C3 * D).to(dhin.dtype)


def _head_tail(U, ssum, ssq, w2, b2, alpha, B, H2, W2, Z, Cmid, Cout, eps):
    cnt = Z * H2 * W2
    mu = (ssum / cnt).view(B, 1, 1, 1, Cmid)
    var = (ssq / cnt).view(B, 1, 1, 1, Cmid) - mu * mu
    rs = torch.rsqrt(var.clamp_min(0) + eps)
    u = U.view(B, H2, W2, Z, Cmid)
    nh = (u - mu) * rs
    a = torch.where(nh > 0, nh, alpha * nh)
    v = a @ w2.t() + b2  # [B,H2,W2,Z,4*Cout]
    v = v.view(B, H2, W2, Z, Cout, 2, 2).permute(0, 4, 3, 1, 5, 2, 6)  # b, co, z, y, i, x, j
    return v.reshape(B, Cout, Z, 2 * H2, 2 * W2), a


def head_out_fwd(U, ssum, ssq, w2, b2, alpha, B, H2, W2, Z, Cmid, Cout, eps=1e-5):
    out, _ = _head_tail(U.float(), ssum, ssq, w2, b2, alpha, B, H2, W2, Z, Cmid, Cout, eps)
    return out.contiguous()


def _head_full(U, w2, b2, alpha, B, H2, W2, Z, Cmid, Cout, eps):
    """InstanceNorm with statistics recomputed from U (so autograd sees the full dependence)."""
    u = U.view(B, H2 * W2 * Z, Cmid)
    ssum, ssq = u.sum(1), (u * u).sum(1)
    return _head_tail(U, ssum, ssq, w2, b2, alpha, B, H2, W2, Z, Cmid, Cout, eps)


@torch.enable_grad()
def head_out_bwd1(U, ssum, ssq, w2, alpha, dout, S1, S2, dalpha, B, H2, W2, Z, Cmid, Cout, eps=1e-5):
    Uf = U.float()
    cnt = Z * H2 * W2
    mu = (ssum / cnt).view(B, 1, 1, 1, Cmid)
    rs = torch.rsqrt(((ssq / cnt).view(B, 1, 1, 1, Cmid) - mu * mu).clamp_min(0) + eps)
    nh = ((Uf.view(B, H2, W2, Z, Cmid) - mu) * rs).detach()
    al = alpha.detach().clone().requires_grad_(True)
    nhr = nh.clone().requires_grad_(True)
    a = torch.where(nhr > 0, nhr, al * nhr)
    dvt = dout.view(B, Cout, Z, H2, 2, W2, 2).permute(0, 3, 5, 2, 1, 4, 6).reshape(B, H2, W2, Z, 4 * Cout)
    dvt = dvt.to(U.dtype).float()
    dA = dvt @ w2
    dn, dal = torch.autograd.grad(a, (nhr, al), dA)
    dalpha += dal
    S1 += dn.sum((1, 2, 3))
    S2 += (dn * nh).sum((1, 2, 3))
    return a.detach().reshape(-1, Cmid).to(U.dtype), dvt.reshape(-1, 4 * Cout).to(U.dtype)


def head_out_bwd1_wgrad(U, ssum, ssq, w2, alpha, dout, S1, S2, dalpha, dW2, db2, B, H2, W2, Z, Cmid, Cout, eps=1e-5):
    """pass 1 + the 1x1x1 convolution's weight / bias gradient (storage-dtype operands, fp32 accumulation)"""
    act, dv = head_out_bwd1(U, ssum, ssq, w2, alpha, dout, S1, S2, dalpha, B, H2, W2, Z, Cmid, Cout, eps)
    dW2.view(4 * Cout, Cmid).add_(dv.float().t() @ act.float())
    db2.add_(dv.float().sum(0))
    return dv


def head_out_bwd2(U, ssum, ssq, w2, alpha, dv, S1, S2, B, H2, W2, Z, Cmid, Cout, eps=1e-5):
    Uf = U.float()
    cnt = Z * H2 * W2
    mu = (ssum / cnt).view(B, 1, 1, 1, Cmid)
    rs = torch.rsqrt(((ssq / cnt).view(B, 1, 1, 1, Cmid) - mu * mu).clamp_min(0) + eps)
    nh = (Uf.view(B, H2, W2, Z, Cmid) - mu) * rs
    dA = dv.float().view(B, H2, W2, Z, 4 * Cout) @ w2
    dn = torch.where(nh > 0, dA, alpha * dA)
    dU = rs * (dn - (S1 / cnt).view(B, 1, 1, 1, Cmid) - nh * (S2 / cnt).view(B, 1, 1, 1, Cmid))
    return dU.reshape(-1, Z * Cmid).to(U.dtype)


@contextlib.contextmanager
def batch():
    """viscy_amd.ops.batch collects the weight-space launches of the block into task lists; here every op runs at once"""
    yield


def flush():
    pass


def batch_open():
    pass


def batch_close():
    pass


def _tap_dst(t, tapmode):
    if tapmode == 0:
        return t
    kx, ky, kz = t % 3, (t // 3) % 3, t // 9
    return (ky * 3 + kx) * 3 + kz


def prep_weight(src, R, Cs, Tn, dtype, *, want=True, want_t=False, gamma=None, tapmode=0):
    w = src.detach().float().reshape(R, Cs, Tn)
    if gamma is not None:
        w = w * gamma.detach().view(1, Cs, 1)
    perm = torch.tensor([_tap_dst(t, tapmode) for t in range(Tn)])
    dst = torch.empty(R, Tn, Cs)
    dst[:, perm, :] = w.permute(0, 2, 1)
    dst = dst.reshape(R, Tn * Cs).to(dtype)
    return (dst if want else None), (dst.t().contiguous() if want_t else None)


def unprep_grad(g, dparam, R, Cs, Tn, *, gamma=None, W=None, dgamma=None, u=None, beta=None, rowsub=None, tapmode=0):
    perm = torch.tensor([_tap_dst(t, tapmode) for t in range(Tn)])
    if rowsub is not None:
        g = g - rowsub.view(R, 1)
    gp = g.view(R, Tn, Cs)[:, perm, :].permute(0, 2, 1)  # [R, Cs, Tn] in parameter order
    if gamma is not None:
        dgamma += (gp * W.detach().reshape(R, Cs, Tn)).sum((0, 2))
        gp = gp * gamma.detach().view(1, Cs, 1)
    if u is not None:
        gp = gp + (u.view(R, 1) * beta.detach().view(1, Cs)).view(R, Cs, 1)
    dparam += gp.reshape(dparam.shape)


def matvec(W, v, b, R, Cc):
    out = W.detach().reshape(R, Cc) @ v.detach()
    return out + b.detach() if b is not None else out


def matvec_t_add(W, u, out, R, Cc):
    out += W.detach().reshape(R, Cc).t() @ u


def transpose_f32(src, dst, A, Bn, accumulate):
    t = src.detach().reshape(A, Bn).t()
    if accumulate:
        dst += t.reshape(dst.shape)
    else:
        dst.copy_(t.reshape(dst.shape))


def prep_head_dgrad(W, Cmid, C3, Zout, dtype):
    Wd = W.detach().float()  # [Cmid, C3, 3, 3, 3]
    dst = torch.zeros(Zout + 2, C3, 3, 3, 3, Cmid)
    for zp in range(Zout + 2):
        zs = min(max(zp - 2, 0), Zout - 3)
        for j in range(3):
            kz = zp - (zs + j)
            if 0 <= kz <= 2:
                # [o, c3, ky, kx] flipped → [c3, ty, tx, o]
                dst[zp, :, :, :, j, :] = Wd[:, :, kz].flip(2, 3).permute(1, 2, 3, 0)
    return dst.reshape((Zout + 2) * C3, 27 * Cmid).to(dtype)


def adamw(p, g, m, v, hyper):
    lr, b1, b2, eps, wd, bc1, bc2, gs = [float(t) for t in hyper.tolist()]
    gr = g * gs
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(gr, alpha=1 - b1)
    v.mul_(b2).addcmul_(gr, gr, value=1 - b2)
    p.addcdiv_(m, v.sqrt() / (bc2**0.5) + eps, value=-lr / bc1)


def mlp_supported(C, hw, M, dtype, mode=None) -> bool:
    """the fused GRN-MLP exists only as a HIP kernel; this backend states the unfused schedule"""
    return False


def fill_(t, value=0.0):
    return t.fill_(value)


def zeros(*shape, device):
    return torch.zeros(shape, dtype=torch.float32, device=device)


def adamw_advance(cfg, step, hyper):
    import math

    c = [float(t) for t in cfg.tolist()]
    t0 = int(step.item())
    lam = 1.0
    if c[6] > 0.5:
        warm, total, mult, cycles = c[7], c[8], c[9], c[10]
        if t0 < warm:
            lam = mult + (1 - mult) * (t0 / max(1.0, warm))
        else:
            lam = max(0.0, 0.5 * (1.0 + math.cos(math.pi * cycles * 2.0 * ((t0 - warm) / max(1.0, total - warm)))))
    t = t0 + 1
    hyper.copy_(torch.tensor([c[0] * lam, c[1], c[2], c[3], c[4], 1 - c[1] ** t, 1 - c[2] ** t, c[5]], dtype=torch.float32))
    step.fill_(t)


def head_conv_supported(H2, W2, c3, cmid, zo, dtype) -> bool:
    """the direct LDS-tiled head convolution exists only as a HIP kernel; the schedule falls back to the z-batched
    implicit GEMMs, which this backend states"""
    return False


def scale_weight_samples(W, s, dtype):
    return (W.detach().float().reshape(W.shape[0], -1)[None, :, :] * s.float()[:, None, :]).to(dtype)


def layer_scale_fold(W, b, gamma):
    R = W.shape[0]
    return (gamma.view(R, 1) * W.reshape(R, -1)).contiguous(), gamma * b


def layer_scale_unfold(dWs, dbs, W, b, gamma, dW, db, dgamma):
    R = W.shape[0]
    dW += (gamma.view(R, 1) * dWs.reshape(R, -1)).view_as(dW)
    db += gamma * dbs
    dgamma += (dWs.reshape(R, -1) * W.reshape(R, -1)).sum(1) + dbs * b


def avgpool_rows_fwd(x, B, hw, C):
    return x.float().view(B, hw, C).mean(1)


def avgpool_rows_bwd(dout, B, hw, C, dtype):
    return (dout / hw).unsqueeze(1).expand(B, hw, C).reshape(B * hw, C).to(dtype)


def bn1d_fwd(x, w, b, rmean, rvar, training, relu, eps=1e-5, momentum=0.1):
    import torch.nn.functional as F

    if training:
        mean, var = x.mean(0), x.var(0, unbiased=False)
    else:
        mean, var = rmean.clone(), rvar.clone()
    y = F.batch_norm(x, rmean, rvar, w, b, training, momentum, eps)
    if relu:
        y = y.relu()
    return y, mean, (var + eps).rsqrt()


def bn1d_bwd(dy, x, y, w, sm, sr, dw, db, training, relu):
    d = dy * (y > 0) if relu else dy
    xh = (x - sm) * sr
    sb, sg = d.sum(0), (d * xh).sum(0)
    dw += sg
    db += sb
    g = w * sr
    B = x.shape[0]
    return g * (d - sb / B - xh * sg / B) if training else g * d


def scale_rows_samples(x, scale, M, C, hw):
    idx = torch.arange(M, device=x.device) // hw
    return (x.float() * scale[idx][:, None]).to(x.dtype)


def rows_select(src, row_map, n_out, C, add=None):
    m = row_map.long()
    out = src[m.clamp_min(0)].clone()
    if add is not None:
        out = (out.float() + add.float()).to(src.dtype)
    out[m < 0] = 0
    return out


def voxel_shuffle_fwd(feat, B, h, w, Cout, D, s, pool):
    import torch.nn.functional as F

    x = feat.float().view(B, h, w, Cout * D * s * s).permute(0, 3, 1, 2)
    x = F.pixel_shuffle(x, s)
    if pool:
        x = F.avg_pool2d(F.pad(x, (s - 1, 0, s - 1, 0)), kernel_size=s, stride=1)
    return x.reshape(B, Cout, D, s * h, s * w).contiguous()


@torch.enable_grad()
def voxel_shuffle_bwd(dout, B, h, w, Cout, D, s, pool, dtype):
    f = torch.zeros(B * h * w, Cout * D * s * s, requires_grad=True)
    voxel_shuffle_fwd(f, B, h, w, Cout, D, s, pool).backward(dout.float())
    return f.grad.to(dtype)
