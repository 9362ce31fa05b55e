"""Tensor-level wrappers over the libvsx C-ABI: PyTorch-ROCm tensors in, tensors out.

PyTorch is used for device memory (caching allocator) and the current stream only; all
arithmetic happens in the HIP kernels.  Every wrapper raises if a tensor is not on the GPU.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Sequence

import torch
from torch import Tensor

from . import _lib as L
from ._lib import VsxGemm, check, dtype_code, lib, ptr, stream


_WS: dict = {}


def _workspace(dev, numel: int) -> Tensor:
    """fp32 scratch the two-stage reductions write their per-block partials into (caller-owned, reused)."""
    w = _WS.get(dev)
    if w is None or w.numel() < numel:
        w = torch.empty(numel, dtype=torch.float32, device=dev)
        _WS[dev] = w
    return w


_DET_WS: dict = {}


def _det(dev, floats: int) -> None:
    """``det_reduce`` flag on: hand the library the scratch for the per-workgroup partial sums of the next launch (grown, never
    freed: a captured hipGraph keeps pointing at it)"""
    if not lib().vsx_get_flag(b"det_reduce"):
        return
    held = _DET_WS.setdefault(dev, [])
    if not held or held[-1].numel() < floats:
        held.append(torch.empty(max(int(floats), 1 << 20), dtype=torch.float32, device=dev))
    check(lib().vsx_det_workspace(ptr(held[-1]), held[-1].numel()), "det_workspace")


def _fill8(arr, vals: Sequence[int] | None):
    if vals:
        for i, v in enumerate(vals):
            arr[i] = int(v)


def gemm(
    kind: str,
    A: Tensor,
    B: Tensor,
    Cout: Tensor,
    M: int,
    N: int,
    K: int,
    lda: int,
    ldb: int,
    ldc: int,
    *,
    dtype: torch.dtype,
    a_mode: int = L.A_ROWS,
    gh: int = 0,
    gw: int = 0,
    cs: int = 0,
    nz: int = 1,
    a_coff: Sequence[int] | None = None,
    b_off: Sequence[int] | None = None,
    c_coff: Sequence[int] | None = None,
    c_mode: int = L.A_ROWS,
    c_cs: int = 0,
    pro: int = L.PRO_NONE,
    grn_s: Tensor | None = None,
    grn_b: Tensor | None = None,
    hw: int = 0,
    epi: int = L.EPI_NONE,
    bias: Tensor | None = None,
    res: Tensor | None = None,
    ldr: int = 0,
    aux: Tensor | None = None,
    ldx: int = 0,
    red0: Tensor | None = None,
    red1: Tensor | None = None,
    colsum: Tensor | None = None,
    C2: Tensor | None = None,
    b_bstride: int = 0,
    rscale: Tensor | None = None,
) -> None:
    """kind='nt': C[M,N] = pro(A)[M,K]·B[N,K]^T (+epilogue);  kind='tn': C[N,K](fp32) += B[M,N]^T·pro(A)[M,K]."""
    for t in (bias, grn_s, grn_b, red0, red1, colsum):
        if t is not None and t.dtype != torch.float32:
            raise TypeError("bias / GRN / reduction buffers must be float32")
    p = VsxGemm()
    p.A, p.B, p.C = ptr(A), ptr(B), ptr(Cout)
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldb, p.ldc = lda, ldb, ldc
    p.a_mode, p.gh, p.gw, p.cs = a_mode, gh, gw, cs
    p.nz = nz
    _fill8(p.a_coff, a_coff)
    _fill8(p.b_off, b_off)
    _fill8(p.c_coff, c_coff)
    p.c_mode, p.c_cs = c_mode, c_cs
    p.pro = pro
    p.grn_s, p.grn_b = ptr(grn_s), ptr(grn_b)
    p.hw = hw
    p.epi = epi
    p.bias, p.res, p.ldr = ptr(bias), ptr(res), ldr
    p.aux, p.ldx = ptr(aux), ldx
    p.red0, p.red1, p.colsum = ptr(red0), ptr(red1), ptr(colsum)
    p.C2 = ptr(C2)
    p.b_bstride = b_bstride
    p.rscale = ptr(rscale)
    if kind == "nt" and epi == L.EPI_BIAS_GELU_SQ:
        _det(A.device, (M // 256 + 1) * N)
    fn = lib().vsx_gemm_nt if kind == "nt" else lib().vsx_gemm_tn
    check(fn(C.byref(p), dtype_code(dtype), stream()), f"gemm_{kind}")


def tn_grn_stats_ok(M: int, N: int, K: int, hw: int, dtype: torch.dtype) -> bool:
    """``gemm("tn", g, dout, dW2, ..., pro=PRO_GRN, aux=W2 (bf16 [N, K]), ldx=K, red0=P)`` — the fc2 weight gradient that also
    delivers the GRN backward statistics P[b, k] = sum_hw dz * g from the per-sample tiles in its accumulators (csrc/gemm.hip,
    gemm_tn_fast_kernel PRO == 2) — serves this shape"""
    return dtype == torch.bfloat16 and hw > 0 and hw % 64 == 0 and M % hw == 0 and N >= 96 and K >= 128 and N % 8 == 0 and K % 8 == 0 \
        and bool(lib().vsx_get_flag(b"tn_rect") & 8)


def dgrad_ln_bwd(dh: Tensor, WT: Tensor, xh: Tensor, rstd: Tensor, M: int, C: int, K: int, mean: Tensor | None = None,
                 out: Tensor | None = None) -> Tensor | None:
    """fc1 data gradient with the block LayerNorm's backward in the GEMM epilogue (VSX_EPI_LN_BWD, csrc/gemm_nt2.hip):
    dy = LN_backward(dh . WT^T; xh, rstd) [M, C] in ONE launch — dx^ is never written.  None if the shape is not served
    (the caller then runs the GEMM and vsx_ln_bwd).  With ``mean``: ``xh`` holds the UN-normalised rows y and ``dh`` the
    row-scaled dh * rstd of mlp_bwd_dh_ln."""
    if dh.dtype != torch.bfloat16 or not lib().vsx_gemm_nt_ln_bwd_supported(M, C, K, dtype_code(dh.dtype)):
        return None
    dy = torch.empty((M, C), dtype=dh.dtype, device=dh.device) if out is None else out
    gemm("nt", dh, WT, dy, M, C, K, K, K, C, dtype=dh.dtype, epi=L.EPI_LN_BWD, aux=xh, ldx=C, grn_s=rstd, grn_b=mean)
    return dy


def gemm_z(kind: str, *args, nz: int, a_coff, b_off, c_coff, **kw) -> None:
    """z-batched GEMM with more than 8 slabs: issue in groups of <= 8."""
    for s in range(0, nz, 8):
        e = min(nz, s + 8)
        gemm(kind, *args, nz=e - s, a_coff=a_coff[s:e], b_off=b_off[s:e], c_coff=c_coff[s:e], **kw)


def ln_fwd(x: Tensor, gamma: Tensor | None, beta: Tensor | None, rows: int, Cc: int, eps: float = 1e-6,
           need_mean: bool = True):
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_mean else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().vsx_ln_fwd(ptr(x), ptr(y), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), rows, Cc, eps,
                           dtype_code(x.dtype), stream()), "ln_fwd")
    return y, mean, rstd


def ln_bwd(dy: Tensor, x: Tensor, mean: Tensor | None, rstd: Tensor, gamma: Tensor | None, add: Tensor | None,
           dgamma: Tensor | None, dbeta: Tensor | None, rows: int, Cc: int) -> Tensor:
    dx = torch.empty_like(dy)
    check(lib().vsx_ln_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(add), ptr(dx), ptr(dgamma),
                           ptr(dbeta), rows, Cc, dtype_code(dy.dtype), stream()), "ln_bwd")
    return dx


def grn_scale(colsq: Tensor, gamma: Tensor, eps: float = 1e-6, out: Tensor | None = None) -> Tensor:
    s = torch.empty_like(colsq) if out is None else out
    check(lib().vsx_grn_scale(ptr(colsq), ptr(gamma), ptr(s), colsq.shape[0], colsq.shape[1], eps, stream()), "grn_scale")
    return s


def grn_bwd_stats(colsq: Tensor, P: Tensor, gamma: Tensor, dgamma: Tensor, eps: float = 1e-6, Sb: Tensor | None = None,
                  dbeta: Tensor | None = None) -> Tensor:
    """Sb [nb, N] = per-sample Σ_hw dz (EPI_DZ red1); dbeta[N] += Σ_b Sb."""
    t = torch.empty_like(colsq)
    rowst = torch.empty_like(colsq)
    if _BATCH is not None:  # the two column reductions into the parameter gradients join the segment's task list
        nb, N = colsq.shape
        check(lib().vsx_grn_bwd_stats(ptr(colsq), ptr(P), None, ptr(gamma), ptr(t), None, None, ptr(rowst), nb, N, eps, stream()),
              "grn_bwd_stats")
        _queue(L.WTASK_REDUCE_ROWS, 0, (nb, N), rowst, dgamma, None, None)
        if Sb is not None:
            _queue(L.WTASK_REDUCE_ROWS, 0, (nb, N), Sb, dbeta, None, None)
        return t
    check(lib().vsx_grn_bwd_stats(ptr(colsq), ptr(P), ptr(Sb), ptr(gamma), ptr(t), ptr(dgamma), ptr(dbeta), ptr(rowst), colsq.shape[0],
                                  colsq.shape[1], eps, stream()), "grn_bwd_stats")
    return t


def grn_gelu_bwd(dz: Tensor, h: Tensor, s: Tensor, t: Tensor, colsum: Tensor, M: int, N: int, hw: int) -> None:
    rows = 2048
    ws = _workspace(dz.device, rows * N)
    check(lib().vsx_grn_gelu_bwd(ptr(dz), ptr(h), ptr(s), ptr(t), ptr(colsum), ptr(ws), rows, M, N, hw,
                                 dtype_code(dz.dtype), stream()), "grn_gelu_bwd")


def dwconv7_fwd(x: Tensor, w: Tensor, bias: Tensor | None, B: int, H: int, W: int, Cc: int) -> Tensor:
    y = torch.empty_like(x)
    check(lib().vsx_dwconv7_fwd(ptr(x), ptr(w), ptr(bias), None, ptr(y), B, H, W, Cc, dtype_code(x.dtype), stream()),
          "dwconv7_fwd")
    return y


def dwconv7_bwd_data(dy: Tensor, w: Tensor, add: Tensor | None, B: int, H: int, W: int, Cc: int) -> Tensor:
    dx = torch.empty_like(dy)
    check(lib().vsx_dwconv7_bwd_data(ptr(dy), ptr(w), ptr(add), ptr(dx), B, H, W, Cc, dtype_code(dy.dtype), stream()),
          "dwconv7_bwd_data")
    return dx


def dwconv7_bwd_weight(dy: Tensor, x: Tensor, dw: Tensor, db: Tensor | None, B: int, H: int, W: int, Cc: int) -> None:
    rows = 256
    ws = _workspace(dy.device, rows * 50 * Cc)
    check(lib().vsx_dwconv7_bwd_weight(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), rows, B, H, W, Cc, dtype_code(dy.dtype),
                                       stream()), "dwconv7_bwd_weight")


def stem_im2col(x: Tensor, kernel: tuple[int, int, int], dtype: torch.dtype, sub: Tensor | None = None,
                div: Tensor | None = None, ld: int | None = None) -> Tensor:
    """patch matrix of the stem; ``ld`` > patch size pads every row with zeros (see vsx_stem_im2col_ld)"""
    if x.dtype != torch.float32:
        raise TypeError("input stacks are float32")
    B, Cin, Z, H, W = x.shape
    kz, ky, kx = kernel
    K = (Z // kz) * Cin * kz * ky * kx
    ld = K if ld is None else ld
    P = torch.empty((B * (H // ky) * (W // kx), ld), dtype=dtype, device=x.device)
    check(lib().vsx_stem_im2col_ld(ptr(x), ptr(P), ptr(sub), ptr(div), B, Cin, Z, H, W, kz, ky, kx, ld, dtype_code(dtype),
                                   stream()), "stem_im2col")
    return P


def im2col3x3(x: Tensor, B: int, H: int, W: int, Cc: int) -> Tensor:
    """[B*H*W, C] channels-last map -> [B*H*W, 9*C] patch matrix of a 3x3 / padding-1 convolution (tap-major columns)"""
    col = torch.empty((B * H * W, 9 * Cc), dtype=x.dtype, device=x.device)
    check(lib().vsx_im2col3x3(ptr(x), ptr(col), B, H, W, Cc, dtype_code(x.dtype), stream()), "im2col3x3")
    return col


def col2im3x3(dcol: Tensor, B: int, H: int, W: int, Cc: int) -> Tensor:
    """transpose of im2col3x3: [B*H*W, 9*C] -> [B*H*W, C]"""
    dx = torch.empty((B * H * W, Cc), dtype=dcol.dtype, device=dcol.device)
    check(lib().vsx_col2im3x3(ptr(dcol), ptr(dx), B, H, W, Cc, dtype_code(dcol.dtype), stream()), "col2im3x3")
    return dx


def pad_cols(src: Tensor, Kp: int) -> Tensor:
    """[R, K] -> [R, Kp] with zero-filled tail columns"""
    R, K = src.shape
    dst = torch.empty((R, Kp), dtype=src.dtype, device=src.device)
    check(lib().vsx_pad_cols(ptr(src), ptr(dst), R, K, Kp, dtype_code(src.dtype), stream()), "pad_cols")
    return dst


def pixel_shuffle_cat_fwd(low: Tensor, skip: Tensor | None, B: int, h: int, w: int, c: int, cs: int) -> Tensor:
    out = torch.empty((B * 4 * h * w, c + cs), dtype=low.dtype, device=low.device)
    check(lib().vsx_pixel_shuffle_cat_fwd(ptr(low), ptr(skip), ptr(out), B, h, w, c, cs, dtype_code(low.dtype), stream()),
          "pixel_shuffle_cat_fwd")
    return out


def pixel_shuffle_cat_bwd(dcat: Tensor, B: int, h: int, w: int, c: int, cs: int):
    dlow = torch.empty((B * h * w, 4 * c), dtype=dcat.dtype, device=dcat.device)
    dskip = torch.empty((B * 4 * h * w, cs), dtype=dcat.dtype, device=dcat.device) if cs else None
    check(lib().vsx_pixel_shuffle_cat_bwd(ptr(dcat), ptr(dlow), ptr(dskip), B, h, w, c, cs, dtype_code(dcat.dtype),
                                          stream()), "pixel_shuffle_cat_bwd")
    return dlow, dskip


def head_shuffle_fwd(dec: Tensor, B: int, h: int, w: int, C3: int, D: int, pool: bool) -> Tensor:
    hin = torch.empty((B * 4 * h * w, C3 * D), dtype=dec.dtype, device=dec.device)
    check(lib().vsx_head_shuffle_fwd(ptr(dec), ptr(hin), B, h, w, C3, D, int(pool), dtype_code(dec.dtype), stream()),
          "head_shuffle_fwd")
    return hin


def head_shuffle_bwd(dhin: Tensor, B: int, h: int, w: int, C3: int, D: int, pool: bool) -> Tensor:
    ddec = torch.empty((B * h * w, 4 * C3 * D), dtype=dhin.dtype, device=dhin.device)
    check(lib().vsx_head_shuffle_bwd(ptr(dhin), ptr(ddec), B, h, w, C3, D, int(pool), dtype_code(dhin.dtype), stream()),
          "head_shuffle_bwd")
    return ddec


def head_out_fwd(U, ssum, ssq, w2, b2, alpha, B, H2, W2, Z, Cmid, Cout, eps=1e-5) -> Tensor:
    out = torch.empty((B, Cout, Z, 2 * H2, 2 * W2), dtype=torch.float32, device=U.device)
    check(lib().vsx_head_out_fwd(ptr(U), ptr(ssum), ptr(ssq), ptr(w2), ptr(b2), ptr(alpha), ptr(out), B, H2, W2, Z, Cmid,
                                 Cout, eps, dtype_code(U.dtype), stream()), "head_out_fwd")
    return out


def head_out_bwd1(U, ssum, ssq, w2, alpha, dout, S1, S2, dalpha, B, H2, W2, Z, Cmid, Cout, eps=1e-5):
    M5 = B * H2 * W2 * Z
    act = torch.empty((M5, Cmid), dtype=U.dtype, device=U.device)
    dv = torch.empty((M5, 4 * Cout), dtype=U.dtype, device=U.device)
    check(lib().vsx_head_out_bwd1(ptr(U), ptr(ssum), ptr(ssq), ptr(w2), ptr(alpha), ptr(dout), ptr(act), ptr(dv), ptr(S1),
                                  ptr(S2), ptr(dalpha), B, H2, W2, Z, Cmid, Cout, eps, dtype_code(U.dtype), stream()),
          "head_out_bwd1")
    return act, dv


def head_out_bwd1_wgrad(U, ssum, ssq, w2, alpha, dout, S1, S2, dalpha, dW2, db2, B, H2, W2, Z, Cmid, Cout, eps=1e-5) -> Tensor:
    """bf16: pass 1 + the 1x1x1 weight / bias gradient (accumulated into dW2 / db2) in one launch; returns dv."""
    M5 = B * H2 * W2 * Z
    dv = torch.empty((M5, 4 * Cout), dtype=U.dtype, device=U.device)
    scratch = torch.empty((B, 4 * Cout * (Cmid + 1)), dtype=torch.float32, device=U.device)
    check(lib().vsx_head_out_bwd1_wgrad(ptr(U), ptr(ssum), ptr(ssq), ptr(w2), ptr(alpha), ptr(dout), ptr(dv), ptr(S1), ptr(S2),
                                        ptr(dalpha), ptr(dW2), ptr(db2), ptr(scratch), B, H2, W2, Z, Cmid, Cout, eps,
                                        dtype_code(U.dtype), stream()), "head_out_bwd1_wgrad")
    return dv


def head_out_bwd2(U, ssum, ssq, w2, alpha, dv, S1, S2, B, H2, W2, Z, Cmid, Cout, eps=1e-5) -> Tensor:
    dU = torch.empty_like(U)
    check(lib().vsx_head_out_bwd2(ptr(U), ptr(ssum), ptr(ssq), ptr(w2), ptr(alpha), ptr(dv), ptr(S1), ptr(S2), ptr(dU), B,
                                  H2, W2, Z, Cmid, Cout, eps, dtype_code(U.dtype), stream()), "head_out_bwd2")
    return dU


# ------------------------------------------------------------------ weight-space task lists
# prep_weight / transpose_f32 / matvec / mlp_pack are tiny (4 - 7 us, the chip idle behind each one) and the step refreshes
# ~200 of them per weight update.  Inside ``with batch():`` they are not launched but collected and handed to
# vsx_weight_tasks when the block ends (or at flush()): a handful of launches.  Nothing inside a batch may consume the output
# of an op queued in the SAME batch — call flush() first.
_BATCH: list | None = None
_BATCH_ON = os.environ.get("VSX_WBATCH", "1") != "0"  # VSX_WBATCH=0: every job its own launch (A/B measurements)
_BATCH_KEEP: list = []  # the queued jobs' tensors stay alive (and their memory un-recycled) until the list is launched
_BATCH_WRITTEN: list = []  # byte ranges [lo, hi) the queued jobs write / read: a later job that overlaps a written range, or writes
_BATCH_READ: list = []     # into a range an earlier job reads, launches the list collected so far first (see _queue)


@contextlib.contextmanager
def batch():
    global _BATCH
    outer = _BATCH
    if _BATCH_ON:
        _BATCH = [] if outer is None else outer
    try:
        yield
    finally:
        if outer is None:
            flush()
            _BATCH = None


def batch_open() -> None:
    """generator-friendly form of ``batch()`` (a context manager must not be held across a yield): start collecting ..."""
    global _BATCH
    if _BATCH is None and _BATCH_ON:
        _BATCH = []


def batch_close() -> None:
    """... launch what was collected and stop collecting"""
    global _BATCH
    if _BATCH is not None:
        flush()
        _BATCH = None


def flush() -> None:
    """launch what the current batch has collected so far (keeps collecting afterwards)"""
    if not _BATCH:
        return
    arr = (L.VsxWTask * len(_BATCH))(*_BATCH)
    del _BATCH[:]
    try:
        check(lib().vsx_weight_tasks(C.addressof(arr), len(arr), stream()), "weight_tasks")
    finally:  # (a failed launch must not leave stale addresses behind for the next list: ADVICE r4)
        del _BATCH_KEEP[:]
        del _BATCH_WRITTEN[:]
        del _BATCH_READ[:]


def _span(t: Tensor):
    """byte range [lo, hi) a (possibly strided) tensor touches"""
    if t.numel() == 0:
        return (t.data_ptr(), t.data_ptr())
    ext = 1 + sum((n - 1) * abs(st) for n, st in zip(t.shape, t.stride()))
    return (t.data_ptr(), t.data_ptr() + ext * t.element_size())


def _overlaps(a, ranges) -> bool:
    return any(a[0] < hi and lo < a[1] for lo, hi in ranges)


def _queue(kind: int, dtype: int, ints, p0, p1, p2, p3, p4=None, p5=None, p6=None, p7=None) -> bool:
    if _BATCH is None:
        return False
    # The jobs of one list run CONCURRENTLY (one launch): a job must not read, write or accumulate into what another job of the
    # same list writes (UNPREP and the accumulating TRANSPOSE / MATVEC_T / REDUCE_ROWS are non-atomic read-modify-writes).
    # Outputs are p1 (+ p2 for PREP / UNPREP); a job that touches memory an already queued job writes — a tied parameter, two
    # finalisers on one gradient, overlapping views of the flat buffers — or WRITES what an already queued job reads launches the
    # list collected so far first (read-after-write, write-after-write and write-after-read, by byte range).
    outs = [t for t in ((p1, p2) if kind in (L.WTASK_PREP, L.WTASK_UNPREP) else (p1,)) if torch.is_tensor(t)]
    ins = [t for t in (p0, p1, p2, p3, p4, p5, p6, p7) if torch.is_tensor(t) and not any(t is o for o in outs)]
    wr, rd = [_span(t) for t in outs], [_span(t) for t in ins]
    if (_BATCH_WRITTEN and any(_overlaps(a, _BATCH_WRITTEN) for a in wr + rd)) or (_BATCH_READ and any(_overlaps(a, _BATCH_READ) for a in wr)):
        flush()
    _BATCH_WRITTEN.extend(wr)
    _BATCH_READ.extend(rd)
    i = list(ints) + [0] * (4 - len(ints))
    _BATCH.append(L.VsxWTask(kind, dtype, i[0], i[1], i[2], i[3], ptr(p0), ptr(p1), ptr(p2), ptr(p3), ptr(p4), ptr(p5), ptr(p6), ptr(p7)))
    _BATCH_KEEP.append((p0, p1, p2, p3, p4, p5, p6, p7))
    return True


def prep_weight(src: Tensor, R: int, Cs: int, Tn: int, dtype: torch.dtype, *, want: bool = True, want_t: bool = False,
                gamma: Tensor | None = None, tapmode: int = 0):
    K = Cs * Tn
    dst = torch.empty((R, K), dtype=dtype, device=src.device) if want else None
    dstT = torch.empty((K, R), dtype=dtype, device=src.device) if want_t else None
    if (dst is not None or dstT is not None) and _queue(L.WTASK_PREP, dtype_code(dtype), (R, Cs, Tn, tapmode), src, dst, dstT, gamma):
        return dst, dstT
    check(lib().vsx_prep_weight(ptr(src), ptr(dst), ptr(dstT), ptr(gamma), R, Cs, Tn, tapmode, dtype_code(dtype), stream()),
          "prep_weight")
    return dst, dstT


def unprep_grad(g: Tensor, dparam: Tensor, R: int, Cs: int, Tn: int, *, gamma=None, W=None, dgamma=None, u=None,
                beta=None, rowsub=None, tapmode=0):
    """``rowsub`` [R]: subtracted from every column of row r of ``g`` first (the rank-1 term of a weight gradient contracted
    with un-centred rows: mlp_bwd_dh_ln)"""
    if _queue(L.WTASK_UNPREP, 0, (R, Cs, Tn, tapmode), g, dparam, dgamma, gamma, W, u, beta, rowsub):
        return
    check(lib().vsx_unprep_grad(ptr(g), ptr(dparam), ptr(gamma), ptr(W), ptr(dgamma), ptr(u), ptr(beta), ptr(rowsub), R, Cs, Tn,
                                tapmode, stream()), "unprep_grad")


def matvec(W: Tensor, v: Tensor, b: Tensor | None, R: int, Cc: int) -> Tensor:
    out = torch.empty(R, dtype=torch.float32, device=W.device)
    if _queue(L.WTASK_MATVEC, 0, (R, Cc), W, out, b, v):
        return out
    check(lib().vsx_matvec(ptr(W), ptr(v), ptr(b), ptr(out), R, Cc, stream()), "matvec")
    return out


def matvec_t_add(W: Tensor, u: Tensor, out: Tensor, R: int, Cc: int) -> None:
    if _queue(L.WTASK_MATVEC_T, 0, (R, Cc), W, out, None, u):
        return
    check(lib().vsx_matvec_t_add(ptr(W), ptr(u), ptr(out), R, Cc, stream()), "matvec_t_add")


def transpose_f32(src: Tensor, dst: Tensor, A: int, Bn: int, accumulate: bool) -> None:
    if _queue(L.WTASK_TRANSPOSE, 0, (A, Bn, int(accumulate)), src, dst, None, None):
        return
    check(lib().vsx_transpose_f32(ptr(src), ptr(dst), A, Bn, int(accumulate), stream()), "transpose_f32")


def prep_head_dgrad(W: Tensor, Cmid: int, C3: int, Zout: int, dtype: torch.dtype) -> Tensor:
    dst = torch.empty(((Zout + 2) * C3, 27 * Cmid), dtype=dtype, device=W.device)
    check(lib().vsx_prep_head_dgrad(ptr(W), ptr(dst), Cmid, C3, Zout, dtype_code(dtype), stream()), "prep_head_dgrad")
    return dst


def fill_(t: Tensor, value: float = 0.0) -> Tensor:
    """t[...] = value for a contiguous fp32 tensor (vsx_fill_f32: a kernel node of this library, not an ATen fill)"""
    if t.dtype != torch.float32:
        raise TypeError("fill_ is for float32 buffers")
    check(lib().vsx_fill_f32(ptr(t), t.numel(), float(value), stream()), "fill_f32")
    return t


def zeros(*shape, device) -> Tensor:
    """zero-filled fp32 tensor (allocation by the caching allocator, fill by vsx_fill_f32)"""
    return fill_(torch.empty(shape, dtype=torch.float32, device=device), 0.0)


def adamw(p: Tensor, g: Tensor, m: Tensor, v: Tensor, hyper: Tensor) -> None:
    check(lib().vsx_adamw(ptr(p), ptr(g), ptr(m), ptr(v), ptr(hyper), p.numel(), stream()), "adamw")


# ------------------------------------------------------------------ fused GRN-MLP (csrc/mlp.hip)
def mlp_supported(C: int, hw: int, M: int, dtype: torch.dtype, mode: int | None = None) -> bool:
    """fused GRN-MLP kernel available: the inference pair (mode None), or one pass — 2 training fc1, 3 backward statistics,
    4 backward dh"""
    if dtype != torch.bfloat16:
        return False
    if mode is None:
        return bool(lib().vsx_mlp_supported(C, hw, M, dtype_code(dtype)))
    return bool(lib().vsx_mlp_mode_supported(C, hw, M, mode, dtype_code(dtype)))


def mlp_pack(W1f: Tensor, W2: Tensor, C: int) -> Tensor:
    """fragment-major LDS image of the prepared fc1 / fc2 weights (bf16 [4C, C] and [C, 4C])"""
    img = torch.empty(int(lib().vsx_mlp_image_bytes(C)), dtype=torch.uint8, device=W1f.device)
    if _queue(L.WTASK_MLP_PACK, 0, (C,), W1f, img, None, W2):
        return img
    check(lib().vsx_mlp_pack(ptr(W1f), ptr(W2), ptr(img), C, stream()), "mlp_pack")
    return img


_GTAB: dict = {}


def _gelu_table(dev) -> Tensor:
    """r(a) = a * Phi(-a) for every bf16 magnitude in [2^-24, 16) (computed once per device, in double precision)"""
    t = _GTAB.get(dev)
    if t is None:
        t = torch.empty(int(lib().vsx_mlp_gelu_table_len()), dtype=torch.float32, device=dev)
        check(lib().vsx_mlp_gelu_table(ptr(t), stream()), "mlp_gelu_table")
        _GTAB[dev] = t
    return t


def mlp_stats(xh: Tensor, img: Tensor, b1: Tensor, colsq: Tensor, M: int, C: int, hw: int, ln_eps: float = 0.0) -> None:
    """colsq[b, 4C] += per-sample column sums of gelu(fc1(xh))^2 — nothing 4C-wide is written.  ``ln_eps`` > 0: ``xh`` holds the
    UN-normalised rows and the kernel applies the block LayerNorm (no affine) in its prologue"""
    _det(xh.device, (M // 256) * 4 * C)
    if ln_eps > 0.0:
        check(lib().vsx_mlp_fwd_ln(ptr(xh), ln_eps, ptr(img), ptr(b1), None, None, None, None, None, None, ptr(colsq),
                                   ptr(_gelu_table(xh.device)), M, C, hw, 0, dtype_code(xh.dtype), stream()), "mlp_stats")
        return
    check(lib().vsx_mlp_fwd(ptr(xh), ptr(img), ptr(b1), None, None, None, None, None, None, ptr(colsq), ptr(_gelu_table(xh.device)),
                            M, C, hw, 0, dtype_code(xh.dtype), stream()), "mlp_stats")


def mlp_fc1(xh: Tensor, img: Tensor, b1: Tensor, colsq: Tensor, M: int, C: int, hw: int, store_h: bool = True):
    """training fc1 on the fused kernel: returns (h, g) [M, 4C] and accumulates the GRN statistics into colsq.  ``store_h=False``
    (where ``mlp_supported(.., 6)``): h is None — the backward recomputes it (mlp_bwd_dh_re)"""
    h = torch.empty((M, 4 * C), dtype=xh.dtype, device=xh.device) if store_h else None
    g = torch.empty((M, 4 * C), dtype=xh.dtype, device=xh.device)
    _det(xh.device, (M // 256) * 4 * C)
    check(lib().vsx_mlp_fc1(ptr(xh), ptr(img), ptr(b1), ptr(colsq), ptr(_gelu_table(xh.device)), ptr(h), ptr(g), M, C, hw,
                            dtype_code(xh.dtype), stream()), "mlp_fc1")
    return h, g


def mlp_fc1_ln(y: Tensor, img: Tensor, b1: Tensor, colsq: Tensor, M: int, C: int, hw: int, eps: float = 1e-6, store_h: bool = True,
               store_xh: bool = True, outs=None):
    """the block LayerNorm (no affine) + training fc1 in one pass over the depthwise convolution's output ``y``: returns
    (xh, rstd, h, g) — what ln_fwd + mlp_fc1 return, without the LayerNorm pass (``store_h=False``: h is None, see mlp_fc1).
    ``store_xh=False``: the normalised rows are not written either; the first item is then the pair (y, mean) the backward
    re-normalises from (mlp_bwd_dh_ln, dgrad_ln_bwd(mean=...))"""
    if outs is not None:  # (mean, rstd, g) views of caller-owned buffers: the sample-chunked schedule (store_h = store_xh = False)
        assert not store_h and not store_xh
        xh, h = None, None
        mean, rstd, g = outs
    else:
        xh = torch.empty_like(y) if store_xh else None
        mean = None if store_xh else torch.empty(M, dtype=torch.float32, device=y.device)
        rstd = torch.empty(M, dtype=torch.float32, device=y.device)
        h = torch.empty((M, 4 * C), dtype=y.dtype, device=y.device) if store_h else None
        g = torch.empty((M, 4 * C), dtype=y.dtype, device=y.device)
    _det(y.device, (M // 256) * 4 * C)
    check(lib().vsx_mlp_fc1_ln(ptr(y), eps, ptr(xh), ptr(rstd), ptr(mean), ptr(img), ptr(b1), ptr(colsq), ptr(_gelu_table(y.device)),
                               ptr(h), ptr(g), M, C, hw, dtype_code(y.dtype), stream()), "mlp_fc1_ln")
    return (xh if store_xh else (y, mean)), rstd, h, g


def grn_q_reduce(Q: Tensor, cs: Tensor, W2: Tensor, s: Tensor, beta: Tensor, P: Tensor, S: Tensor, dW2: Tensor, db2: Tensor) -> None:
    """GRN statistics P, S and the fc2 weight / bias gradient from the per-sample products Q[b] = dout_b^T g_b and the per-sample
    column sums cs[b] of dout (see vsx_grn_q_reduce)"""
    nb, C = cs.shape
    nws = int(lib().vsx_grn_q_reduce_ws_floats(nb, C))
    ws = _workspace(Q.device, nws)
    check(lib().vsx_grn_q_reduce(ptr(Q), ptr(cs), ptr(W2), ptr(s), ptr(beta), ptr(P), ptr(S), ptr(dW2), ptr(db2), ptr(ws), nws,
                                 nb, C, dtype_code(W2.dtype), stream()), "grn_q_reduce")


def mlp_bwd_stats(dout: Tensor, img2: Tensor, g: Tensor, P: Tensor, S: Tensor, M: int, C: int, hw: int) -> None:
    """P[b, 4C] += sum_hw dz * g, S[b, 4C] += sum_hw dz with dz = dout . W2 recomputed on chip (img2 = mlp_pack(W2T, ...))"""
    check(lib().vsx_mlp_bwd_stats(ptr(dout), ptr(img2), ptr(g), ptr(P), ptr(S), M, C, hw, dtype_code(dout.dtype), stream()),
          "mlp_bwd_stats")


def mlp_bwd_dh(dout: Tensor, img2: Tensor, h: Tensor, s: Tensor, t: Tensor, colsum: Tensor, M: int, C: int, hw: int) -> Tensor:
    """dh = (dz * s + gelu(h) * t) * gelu'(h), dz recomputed; colsum[4C] += column sums of dh"""
    dh = torch.empty((M, 4 * C), dtype=dout.dtype, device=dout.device)
    rows = M // int(lib().vsx_mlp_rows_per_workgroup(C, hw, M))
    ws = _workspace(dout.device, rows * 4 * C)
    check(lib().vsx_mlp_bwd_dh(ptr(dout), ptr(img2), ptr(h), ptr(s), ptr(t), ptr(dh), ptr(ws), rows, ptr(colsum),
                               ptr(_gelu_table(dout.device)), M, C, hw, dtype_code(dout.dtype), stream()), "mlp_bwd_dh")
    return dh


def mlp_bwd_dh_re(dout: Tensor, xh: Tensor, img2: Tensor, img: Tensor, b1: Tensor, s: Tensor, t: Tensor, colsum: Tensor, M: int, C: int,
                  hw: int) -> Tensor:
    """mlp_bwd_dh without a stored pre-activation: h = xh . W1'^T + b1 is recomputed on chip from the normalised rows (img = the
    forward image mlp_pack(W1f, W2), b1 = the folded fc1 bias: bit-identical to what mlp_fc1 would have stored)"""
    dh = torch.empty((M, 4 * C), dtype=dout.dtype, device=dout.device)
    rows = M // int(lib().vsx_mlp_rows_per_workgroup(C, hw, M))
    ws = _workspace(dout.device, rows * 4 * C)
    check(lib().vsx_mlp_bwd_dh_re(ptr(dout), ptr(xh), ptr(img2), ptr(img), ptr(b1), ptr(s), ptr(t), ptr(dh), ptr(ws), rows, ptr(colsum),
                                  ptr(_gelu_table(dout.device)), M, C, hw, dtype_code(dout.dtype), stream()), "mlp_bwd_dh_re")
    return dh


def mlp_bwd_dh_ln(dout: Tensor, y: Tensor, mean: Tensor, rstd: Tensor, img2: Tensor, img: Tensor, b1: Tensor, s: Tensor, t: Tensor,
                  colsum2: Tensor, M: int, C: int, hw: int, out: Tensor | None = None) -> Tensor:
    """mlp_bwd_dh_re for a block that stored no normalised rows: x^ is re-formed from the LayerNorm input ``y`` and its row
    statistics; returns dh' = dh * rstd (row-scaled); colsum2 [2, 4C] += {column sums of the UNSCALED dh, u = sum_r dh' * mean}
    (see vsx_mlp_bwd_dh_ln for what the consumers do with them).  ``out``: a caller-owned [M, 4C] buffer (sample-chunked schedule)"""
    dh = torch.empty((M, 4 * C), dtype=dout.dtype, device=dout.device) if out is None else out
    rows = M // int(lib().vsx_mlp_rows_per_workgroup(C, hw, M))
    ws = _workspace(dout.device, rows * 8 * C)
    check(lib().vsx_mlp_bwd_dh_ln(ptr(dout), ptr(y), ptr(mean), ptr(rstd), ptr(img2), ptr(img), ptr(b1), ptr(s), ptr(t), ptr(dh), ptr(ws),
                                  rows, ptr(colsum2), ptr(_gelu_table(dout.device)), M, C, hw, dtype_code(dout.dtype), stream()),
          "mlp_bwd_dh_ln")
    return dh


def mlp_out(xh: Tensor, img: Tensor, b1: Tensor, s: Tensor, beta: Tensor, b2: Tensor, res: Tensor, rscale: Tensor | None,
            M: int, C: int, hw: int, ln_eps: float = 0.0) -> Tensor:
    """out = res + rscale * (fc2(gelu(fc1(xh)) * s + beta) + b2), hidden activation kept on chip (``ln_eps``: see mlp_stats)"""
    out = torch.empty((M, C), dtype=xh.dtype, device=xh.device)
    if ln_eps > 0.0:
        check(lib().vsx_mlp_fwd_ln(ptr(xh), ln_eps, ptr(img), ptr(b1), ptr(s), ptr(beta), ptr(b2), ptr(res), ptr(rscale), ptr(out),
                                   None, ptr(_gelu_table(xh.device)), M, C, hw, 1, dtype_code(xh.dtype), stream()), "mlp_out")
        return out
    check(lib().vsx_mlp_fwd(ptr(xh), ptr(img), ptr(b1), ptr(s), ptr(beta), ptr(b2), ptr(res), ptr(rscale), ptr(out), None,
                            ptr(_gelu_table(xh.device)), M, C, hw, 1, dtype_code(xh.dtype), stream()), "mlp_out")
    return out


def adamw_advance(cfg: Tensor, step: Tensor, hyper: Tensor) -> None:
    """schedule / bias corrections of the next optimiser step from device state (csrc/optim.hip): no host memory involved"""
    if step.dtype != torch.int32 or cfg.dtype != torch.float64:
        raise TypeError("the optimiser step counter is int32, the schedule constants float64")
    check(lib().vsx_adamw_advance(ptr(cfg), ptr(step), ptr(hyper), stream()), "adamw_advance")


# ------------------------------------------------------------------ direct head convolution (csrc/headconv.hip)
def head_conv_supported(H2: int, W2: int, c3: int, cmid: int, zo: int, dtype: torch.dtype) -> bool:
    return bool(lib().vsx_head_conv_supported(H2, W2, c3, cmid, zo, dtype_code(dtype)))


def head_conv_fwd(hin: Tensor, Wc: Tensor, bias: Tensor | None, ssum: Tensor, ssq: Tensor, B: int, H2: int, W2: int, c3: int,
                  cmid: int, zo: int) -> Tensor:
    U = torch.empty((B * H2 * W2, zo * cmid), dtype=hin.dtype, device=hin.device)
    _det(hin.device, B * (H2 // 8) * (W2 // 16) * 64)  # one row of 64 partials per 16 x 8-pixel workgroup
    check(lib().vsx_head_conv_fwd(ptr(hin), ptr(Wc), ptr(bias), ptr(U), ptr(ssum), ptr(ssq), B, H2, W2, c3, cmid, zo,
                                  dtype_code(hin.dtype), stream()), "head_conv_fwd")
    return U


def head_conv_wgrad(hin: Tensor, dU: Tensor, dW: Tensor, db: Tensor | None, B: int, H2: int, W2: int, c3: int, cmid: int,
                    zo: int) -> None:
    check(lib().vsx_head_conv_wgrad(ptr(hin), ptr(dU), ptr(dW), ptr(db), B, H2, W2, c3, cmid, zo, dtype_code(hin.dtype),
                                    stream()), "head_conv_wgrad")


def head_conv_dgrad_prep(Wc: Tensor) -> Tensor:
    Wp = torch.empty(45 * 2 * 64 * 8, dtype=Wc.dtype, device=Wc.device)
    check(lib().vsx_head_conv_dgrad_prep(ptr(Wc), ptr(Wp), dtype_code(Wc.dtype), stream()), "head_conv_dgrad_prep")
    return Wp


def head_conv_dgrad(dU: Tensor, Wp: Tensor, B: int, H2: int, W2: int, c3: int, cmid: int, zo: int) -> Tensor:
    dhin = torch.empty((B * H2 * W2, (zo + 2) * c3), dtype=dU.dtype, device=dU.device)
    check(lib().vsx_head_conv_dgrad(ptr(dU), ptr(Wp), ptr(dhin), B, H2, W2, c3, cmid, zo, dtype_code(dU.dtype), stream()),
          "head_conv_dgrad")
    return dhin


def scale_weight_samples(W: Tensor, s: Tensor, dtype: torch.dtype) -> Tensor:
    """out[b] = dtype(W * s[b][None, :]) — the GRN scale folded into the fc2 weights, one matrix per batch sample"""
    R = W.shape[0]
    K = W.numel() // R  # nn.Linear [R, K] or 1x1 nn.Conv2d [R, K, 1, 1] (timm conv_mlp backbones)
    B = s.shape[0]
    out = torch.empty((B, R, K), dtype=dtype, device=W.device)
    check(lib().vsx_scale_weight_samples(ptr(W), ptr(s), ptr(out), B, R, K, dtype_code(dtype), stream()), "scale_weight_samples")
    return out


def layer_scale_fold(W: Tensor, b: Tensor, gamma: Tensor):
    R = W.shape[0]
    K = W.numel() // R
    Ws, bs = torch.empty((R, K), dtype=torch.float32, device=W.device), torch.empty(R, dtype=torch.float32, device=W.device)
    check(lib().vsx_layer_scale_fold(ptr(W), ptr(b), ptr(gamma), ptr(Ws), ptr(bs), R, K, stream()), "layer_scale_fold")
    return Ws, bs


def layer_scale_unfold(dWs: Tensor, dbs: Tensor, W: Tensor, b: Tensor, gamma: Tensor, dW: Tensor, db: Tensor, dgamma: Tensor) -> None:
    R = W.shape[0]
    K = W.numel() // R
    check(lib().vsx_layer_scale_unfold(ptr(dWs), ptr(dbs), ptr(W), ptr(b), ptr(gamma), ptr(dW), ptr(db), ptr(dgamma), R, K, stream()),
          "layer_scale_unfold")


def avgpool_rows_fwd(x: Tensor, B: int, hw: int, C: int) -> Tensor:
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    check(lib().vsx_avgpool_rows_fwd(ptr(x), ptr(out), B, hw, C, dtype_code(x.dtype), stream()), "avgpool_rows_fwd")
    return out


def avgpool_rows_bwd(dout: Tensor, B: int, hw: int, C: int, dtype: torch.dtype) -> Tensor:
    dx = torch.empty((B * hw, C), dtype=dtype, device=dout.device)
    check(lib().vsx_avgpool_rows_bwd(ptr(dout), ptr(dx), B, hw, C, dtype_code(dtype), stream()), "avgpool_rows_bwd")
    return dx


def bn1d_fwd(x: Tensor, w: Tensor, b: Tensor, rmean: Tensor, rvar: Tensor, training: bool, relu: bool, eps: float = 1e-5,
             momentum: float = 0.1):
    B, F = x.shape
    y = torch.empty_like(x)
    sm = torch.empty(F, dtype=torch.float32, device=x.device)
    sr = torch.empty(F, dtype=torch.float32, device=x.device)
    check(lib().vsx_bn1d_fwd(ptr(x), ptr(w), ptr(b), ptr(rmean), ptr(rvar), ptr(y), ptr(sm), ptr(sr), B, F, eps, momentum,
                             int(training), int(relu), stream()), "bn1d_fwd")
    return y, sm, sr


def bn1d_bwd(dy: Tensor, x: Tensor, y: Tensor, w: Tensor, sm: Tensor, sr: Tensor, dw: Tensor, db: Tensor, training: bool,
             relu: bool) -> Tensor:
    B, F = x.shape
    dx = torch.empty_like(x)
    check(lib().vsx_bn1d_bwd(ptr(dy), ptr(x), ptr(y), ptr(w), ptr(sm), ptr(sr), ptr(dx), ptr(dw), ptr(db), B, F, int(training),
                             int(relu), stream()), "bn1d_bwd")
    return dx


def ntxent_fwd(E: Tensor, labels: Tensor, temperature: float, beta: float):
    N, D = E.shape
    dev = E.device
    En, inv = torch.empty_like(E), torch.empty(N, dtype=torch.float32, device=dev)
    S, dS = torch.empty((N, N), dtype=torch.float32, device=dev), torch.empty((N, N), dtype=torch.float32, device=dev)
    rows, acc = torch.empty(2 * N, dtype=torch.float32, device=dev), torch.empty(2, dtype=torch.float32, device=dev)
    check(lib().vsx_ntxent_fwd(ptr(E), ptr(labels), ptr(En), ptr(inv), ptr(S), ptr(dS), ptr(rows), ptr(acc), N, D,
                               float(temperature), float(beta), stream()), "ntxent_fwd")
    return acc, (dS, En, inv)


def ntxent_bwd(saved, acc: Tensor, gout: Tensor) -> Tensor:
    dS, En, inv = saved
    N, D = En.shape
    dE = torch.empty_like(En)
    check(lib().vsx_ntxent_bwd(ptr(dS), ptr(En), ptr(inv), ptr(acc), ptr(gout), ptr(dE), N, D, stream()), "ntxent_bwd")
    return dE


def scale_rows_samples(x: Tensor, scale: Tensor, M: int, C: int, hw: int) -> Tensor:
    out = torch.empty_like(x)
    check(lib().vsx_scale_rows_samples(ptr(x), ptr(scale), ptr(out), M, C, hw, dtype_code(x.dtype), stream()), "scale_rows_samples")
    return out


def rows_select(src: Tensor, row_map: Tensor, n_out: int, C: int, add: Tensor | None = None) -> Tensor:
    """dst[r] = map[r] >= 0 ? src[map[r]] (+ add[r]) : 0 — masked_patchify / masked_unpatchify / mask multiply of fcmae.py:95-141."""
    assert row_map.dtype == torch.int32 and row_map.numel() == n_out
    dst = torch.empty((n_out, C), dtype=src.dtype, device=src.device)
    check(lib().vsx_rows_select(ptr(src), ptr(row_map), ptr(add) if add is not None else None, ptr(dst), n_out, C,
                                dtype_code(src.dtype), stream()), "rows_select")
    return dst


def masked_mse_fwd(pred: Tensor, orig: Tensor, mask_u8: Tensor):
    B, C, Z, H, W = pred.shape
    acc = torch.empty(2, dtype=torch.float32, device=pred.device)
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    check(lib().vsx_masked_mse_fwd(ptr(pred), ptr(orig), ptr(mask_u8), ptr(acc), ptr(loss), B, C, Z, H * W, stream()), "masked_mse_fwd")
    return loss, acc


def masked_mse_bwd(pred: Tensor, orig: Tensor, mask_u8: Tensor, acc: Tensor, gout: Tensor) -> Tensor:
    B, C, Z, H, W = pred.shape
    dpred = torch.empty_like(pred)
    check(lib().vsx_masked_mse_bwd(ptr(pred), ptr(orig), ptr(mask_u8), ptr(acc), ptr(gout), ptr(dpred), B, C, Z, H * W, stream()),
          "masked_mse_bwd")
    return dpred


def voxel_shuffle_fwd(feat: Tensor, B: int, h: int, w: int, Cout: int, D: int, s: int, pool: bool) -> Tensor:
    out = torch.empty((B, Cout, D, s * h, s * w), dtype=torch.float32, device=feat.device)
    check(lib().vsx_voxel_shuffle_fwd(ptr(feat), ptr(out), B, h, w, Cout, D, s, int(pool), dtype_code(feat.dtype), stream()),
          "voxel_shuffle_fwd")
    return out


def voxel_shuffle_bwd(dout: Tensor, B: int, h: int, w: int, Cout: int, D: int, s: int, pool: bool, dtype: torch.dtype) -> Tensor:
    dfeat = torch.empty((B * h * w, Cout * D * s * s), dtype=dtype, device=dout.device)
    check(lib().vsx_voxel_shuffle_bwd(ptr(dout), ptr(dfeat), B, h, w, Cout, D, s, int(pool), dtype_code(dtype), stream()),
          "voxel_shuffle_bwd")
    return dfeat
