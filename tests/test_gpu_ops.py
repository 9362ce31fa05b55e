"""GPU: every HIP op (through the C-ABI, viscy_amd.ops) against its plain-PyTorch statement
(tests/ref_ops.py) on identical seeded inputs, in fp32 (parity mode) and bf16 (production mode).

fp32 tolerance: 1e-3 relative to the tensor's max magnitude (the north-star bar);
bf16 tolerance: 2e-2 (storage rounding of operands/outputs is shared by both sides, the residual
is accumulation order).
"""

import os

import pytest
import torch

from tests import ref_ops as R
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


SELF_CHECK = bool(os.environ.get("VSX_TEST_SELF"))  # harness self-check on CPU: reference vs reference
DEV = "cpu" if SELF_CHECK else "cuda"


def _hip():
    if SELF_CHECK:
        return R
    from viscy_amd import ops

    return ops


def tol(dt):
    return 1e-3 if dt == torch.float32 else 2e-2


def close(a, b, dt, what="", scale=None):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    s = scale if scale is not None else b.abs().max().clamp_min(1e-6).item()
    err = (a - b).abs().max().item() / s
    assert err <= tol(dt), f"{what}: max err {err:.3e} (scale {s:.3e}) > {tol(dt)}"
    return err


def rnd(*shape, dt=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt)


def cuda(*ts):
    return [t.to(DEV) if t is not None else None for t in ts]


# ------------------------------------------------------------------ ABI / library
def test_library_loads_and_reports_errors():
    if SELF_CHECK:
        pytest.skip("self-check")
    from viscy_amd import _lib

    l = _lib.lib()
    assert l.vsx_version() >= 1
    with pytest.raises(RuntimeError, match="bad arguments|null"):
        _lib.check(l.vsx_adamw(None, None, None, None, None, 0, None), "adamw")


# ------------------------------------------------------------------ GEMM nt
GEMM_CASES = [
    # M, N, K, hw
    (256, 128, 96, 256),
    (200, 96, 80, 100),     # ragged M, tile spans 2 samples
    (8, 384, 40, 4),        # tiny feature map: many samples per tile
    (640, 224, 144, 320),
    (384, 32, 216, 384),
    (130, 8, 64, 130),
    (512, 384, 256, 256),   # lean instantiation, 128-byte K slabs (BK = 64, single LDS buffer), 3 N tiles
    (1024, 224, 512, 512),  # lean, BK = 64, ragged last N tile
    (384, 192, 224, 128),   # lean, BK = 32 (K % 64 != 0)
    (448, 384, 256, 64),    # lean with TWO samples per 128-row tile (8 x 8 feature maps), odd sample count, BK = 64
    (512, 192, 96, 64),     # the same, BK = 32
]


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("M,N,K,hw", GEMM_CASES)
@pytest.mark.parametrize("epi", [R.EPI_NONE, R.EPI_BIAS, R.EPI_BIAS_GELU_SQ, R.EPI_BIAS_RES, R.EPI_DZ, R.EPI_BIAS_STATS])
def test_gemm_nt_rows(dt, M, N, K, hw, epi):
    H = _hip()
    nb = (M + hw - 1) // hw
    A = rnd(M, K, dt=dt, seed=1)
    Bw = rnd(N, K, dt=dt, seed=2, scale=K**-0.5)
    bias = rnd(N, seed=3)
    res = rnd(M, N, dt=dt, seed=4)
    aux = rnd(M, N, dt=dt, seed=5)
    kw = dict(dtype=dt, hw=hw, epi=epi)
    if epi in (R.EPI_BIAS, R.EPI_BIAS_GELU_SQ, R.EPI_BIAS_RES, R.EPI_BIAS_STATS):
        kw["bias"] = bias
    if epi == R.EPI_BIAS_RES:
        kw.update(res=res, ldr=N)
    if epi == R.EPI_DZ:
        kw.update(aux=aux, ldx=N)

    def run(ops, dev):
        k2 = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
        C = torch.zeros(M, N, dtype=dt, device=dev)
        r0 = torch.zeros(nb, N, device=dev)
        r1 = torch.zeros(nb, N, device=dev)
        if epi in (R.EPI_BIAS_GELU_SQ, R.EPI_DZ, R.EPI_BIAS_STATS):
            k2.update(red0=r0, red1=r1)
        C2 = torch.zeros(M, N, dtype=dt, device=dev)
        if epi == R.EPI_BIAS_GELU_SQ:
            k2.update(C2=C2)
        ops.gemm("nt", A.to(dev), Bw.to(dev), C, M, N, K, K, K, N, **k2)
        return C, r0, r1, C2

    Cr, r0r, r1r, C2r = run(R, "cpu")
    Cg, r0g, r1g, C2g = run(H, DEV)
    close(Cg, Cr, dt, "C")
    close(C2g, C2r, dt, "C2", scale=1.0)
    if epi in (R.EPI_BIAS_GELU_SQ, R.EPI_DZ, R.EPI_BIAS_STATS):
        close(r0g, r0r, dt, "red0")
    if epi in (R.EPI_DZ, R.EPI_BIAS_STATS):
        close(r1g, r1r, dt, "red1")


NT2_CASES = [
    # M, N, K, hw  (second-generation NT kernel, csrc/gemm_nt2.hip: 256 x 128 tiles, LDS-DMA operands, wave-private epilogue)
    (512, 128, 96, 256),
    (768, 224, 512, 256),    # ragged last N tile
    (512, 96, 384, 512),     # one partial N tile
    (1024, 384, 256, 64),    # four samples per tile (8 x 8 feature maps): per-wave statistics
    (512, 192, 96, 128),     # two samples per tile
    (256, 1536, 384, 256),   # the C = 384 fc1 / dz shape of one sample: four 384-wide column tiles
    (1024, 384, 1536, 1024), # one 384-wide tile
    (512, 3072, 96, 64),     # eight 384-wide tiles, four samples per tile
    (512, 448, 64, 256),     # 384 + ragged 64: two 256-wide tiles
    (256, 896, 224, 256),    # 256-wide tiles, ragged last (128 used)
]


@pytest.mark.parametrize("M,N,K,hw", NT2_CASES)
@pytest.mark.parametrize("epi", [R.EPI_NONE, R.EPI_BIAS, R.EPI_BIAS_GELU_SQ, R.EPI_BIAS_RES, R.EPI_DZ])
def test_gemm_nt2_kernel(M, N, K, hw, epi):
    """the second-generation NT kernel (forced on for every tile count) against the plain-PyTorch statement AND against the
    first-generation kernels on the same inputs (bit-identical outputs: same MFMA order per output element)"""
    if SELF_CHECK:
        pytest.skip("self-check")
    from viscy_amd import _lib

    H, dt = _hip(), torch.bfloat16
    nb = M // hw
    A = rnd(M, K, dt=dt, seed=1)
    Bw = rnd(N, K, dt=dt, seed=2, scale=K**-0.5)
    bias, res, aux = rnd(N, seed=3), rnd(M, N, dt=dt, seed=4), rnd(M, N, dt=dt, seed=5)
    rscale = (torch.arange(nb) % 2).float() * 1.25
    kw = dict(dtype=dt, hw=hw, epi=epi)
    if epi in (R.EPI_BIAS, R.EPI_BIAS_GELU_SQ, R.EPI_BIAS_RES):
        kw["bias"] = bias
    if epi == R.EPI_BIAS_RES:
        kw.update(res=res, ldr=N, rscale=rscale)
    if epi == R.EPI_DZ:
        kw.update(aux=aux, ldx=N)

    def run(ops, dev):
        k2 = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
        C = torch.full((M, N), float("nan"), dtype=dt, device=dev)
        r0, r1 = torch.zeros(nb, N, device=dev), torch.zeros(nb, N, device=dev)
        if epi in (R.EPI_BIAS_GELU_SQ, R.EPI_DZ):
            k2.update(red0=r0, red1=r1)
        C2 = torch.zeros(M, N, dtype=dt, device=dev)
        if epi == R.EPI_BIAS_GELU_SQ:
            k2.update(C2=C2)
        ops.gemm("nt", A.to(dev), Bw.to(dev), C, M, N, K, K, K, N, **k2)
        return C, r0, r1, C2

    l = _lib.lib()
    old = l.vsx_get_flag(b"nt2")
    try:
        l.vsx_set_flag(b"nt2", 3)
        Cg, r0g, r1g, C2g = run(H, DEV)
        l.vsx_set_flag(b"nt2", 0)
        C1, r01, r11, C21 = run(H, DEV)
    finally:
        l.vsx_set_flag(b"nt2", old)
    Cr, r0r, r1r, C2r = run(R, "cpu")
    close(Cg, Cr, dt, "C")
    close(C2g, C2r, dt, "C2", scale=1.0)
    if epi in (R.EPI_BIAS_GELU_SQ, R.EPI_DZ):
        close(r0g, r0r, dt, "red0")
        torch.testing.assert_close(r0g, r01, rtol=1e-4, atol=1e-3 * float(r01.abs().max()))
    if epi == R.EPI_DZ:
        close(r1g, r1r, dt, "red1")
    assert torch.equal(Cg, C1), "first- and second-generation kernels differ"
    assert torch.equal(C2g, C21)


@pytest.mark.parametrize("M,N,K,hw", [(512, 384, 1536, 256), (768, 224, 512, 256), (512, 96, 384, 512), (256, 768, 3072, 256)])
def test_gemm_nt2_grn_prologue(M, N, K, hw):
    """the fc2 forward with the GRN prologue on the second-generation kernel (A fragments scaled in registers): bit-identical
    to the first-generation prologue kernel, and equal to the plain-PyTorch statement"""
    if SELF_CHECK:
        pytest.skip("self-check")
    from viscy_amd import _lib

    H, dt = _hip(), torch.bfloat16
    A, Bw = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, seed=2, scale=K**-0.5)
    s, beta = 1 + 0.3 * rnd(M // hw, K, seed=3), 0.1 * rnd(K, seed=4)
    res, bias = rnd(M, N, dt=dt, seed=5), rnd(N, seed=6)

    def run(ops, dev):
        C = torch.full((M, N), float("nan"), dtype=dt, device=dev)
        ops.gemm("nt", A.to(dev), Bw.to(dev), C, M, N, K, K, K, N, dtype=dt, pro=R.PRO_GRN, grn_s=s.to(dev), grn_b=beta.to(dev), hw=hw,
                 epi=R.EPI_BIAS_RES, bias=bias.to(dev), res=res.to(dev), ldr=N)
        return C

    l = _lib.lib()
    old = l.vsx_get_flag(b"nt2")
    try:
        l.vsx_set_flag(b"nt2", 3)
        C2 = run(H, DEV)
        l.vsx_set_flag(b"nt2", 0)
        C1 = run(H, DEV)
    finally:
        l.vsx_set_flag(b"nt2", old)
    close(C2, run(R, "cpu"), dt, "grn-prologue gemm (second generation)")
    assert torch.equal(C2, C1)


@pytest.mark.parametrize("M,C", [(512, 96), (768, 192), (512, 224), (256, 256), (1024, 64)])
def test_fc1_data_gradient_with_layernorm_backward_epilogue(M, C):
    """VSX_EPI_LN_BWD (csrc/gemm_nt2.hip): dy = LN_backward(dh . W1; xh, rstd) in one launch against the unfused pair (NT GEMM
    that stores dx^ in bf16, then vsx_ln_bwd) and against the plain-PyTorch statement with the same rounding point"""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import ops

    dt, K = torch.bfloat16, 4 * C
    dh = rnd(M, K, dt=dt, seed=1).cuda()
    WT = rnd(C, K, dt=dt, seed=2, scale=K**-0.5).cuda()
    y = rnd(M, C, seed=3) * (1 + torch.arange(M).float()[:, None] / M) + 0.3
    mu, var = y.mean(1, keepdim=True), y.var(1, unbiased=False, keepdim=True)
    rstd = (var + 1e-6).rsqrt()
    xh = ((y - mu) * rstd).to(dt).cuda()
    rstd = rstd[:, 0].contiguous().cuda()
    assert ops.dgrad_ln_bwd(dh, WT, xh, rstd, M + 8, C, K) is None  # M % 256 != 0: not served, the caller falls back
    dy = ops.dgrad_ln_bwd(dh, WT, xh, rstd, M, C, K)
    assert dy is not None and dy.shape == (M, C) and dy.dtype == dt
    dxh = torch.empty((M, C), dtype=dt, device="cuda")
    ops.gemm("nt", dh, WT, dxh, M, C, K, K, K, C, dtype=dt)
    dy_unfused = ops.ln_bwd(dxh, xh, None, rstd, None, None, None, None, M, C)
    d = (dh.float() @ WT.float().T).to(dt).float()   # dx^ as the unfused pair stores it
    xf = xh.float()
    ref = rstd[:, None] * (d - d.mean(1, keepdim=True) - xf * (d * xf).mean(1, keepdim=True))
    close(dy, ref, dt, "fused vs statement")
    close(dy, dy_unfused, dt, "fused vs unfused pair")


def test_gemm_nt2_per_sample_weights():
    """VsxGemm.b_bstride on the second-generation kernel (the fc2 of the large feature maps)"""
    if SELF_CHECK:
        pytest.skip("self-check")
    from viscy_amd import _lib

    H, dt = _hip(), torch.bfloat16
    M, N, K, hw = 1024, 96, 384, 256
    nb = M // hw
    A = rnd(M, K, dt=dt, seed=1).to(DEV)
    Ws = rnd(nb, N, K, dt=dt, seed=2, scale=K**-0.5).to(DEV)
    res, bias = rnd(M, N, dt=dt, seed=5).to(DEV), rnd(N, seed=6).to(DEV)
    l = _lib.lib()
    old = l.vsx_get_flag(b"nt2")
    outs = []
    try:
        for flag in (3, 0):
            l.vsx_set_flag(b"nt2", flag)
            C = torch.zeros(M, N, dtype=dt, device=DEV)
            H.gemm("nt", A, Ws, C, M, N, K, K, K, N, dtype=dt, hw=hw, b_bstride=N * K, epi=R.EPI_BIAS_RES, bias=bias, res=res, ldr=N)
            outs.append(C)
    finally:
        l.vsx_set_flag(b"nt2", old)
    ref = torch.cat([A[b * hw:(b + 1) * hw].float() @ Ws[b].float().T for b in range(nb)]) + bias + res.float()
    close(outs[0], ref, dt, "per-sample weights")
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("M,N,K,hw", [(200, 40, 48, 100), (512, 384, 256, 256), (384, 192, 224, 128)],
                         ids=["generic", "lean_bk64", "lean_bk32"])
def test_gemm_nt_gelu_sq_without_preactivation_store(dt, M, N, K, hw):
    """inference: C = NULL with EPI_BIAS_GELU_SQ keeps only the activation; bit-identical C2 / same statistics, C untouched"""
    H = _hip()
    nb = (M + hw - 1) // hw
    A, Bw, bias = rnd(M, K, dt=dt, seed=1).to(DEV), rnd(N, K, dt=dt, seed=2, scale=K**-0.5).to(DEV), rnd(N, seed=3).to(DEV)
    outs = []
    for keep in (True, False):
        C = torch.zeros(M, N, dtype=dt, device=DEV) if keep else None
        C2 = torch.zeros(M, N, dtype=dt, device=DEV)
        r0 = torch.zeros(nb, N, device=DEV)
        H.gemm("nt", A, Bw, C, M, N, K, K, K, N, dtype=dt, hw=hw, epi=R.EPI_BIAS_GELU_SQ, bias=bias, red0=r0, C2=C2)
        outs.append((C2, r0))
    assert torch.equal(outs[0][0], outs[1][0])
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("M,N,K,hw", [(192, 40, 160, 64), (768, 224, 384, 256), (512, 128, 96, 128), (448, 768, 384, 64),
                                      (256, 192, 96, 64)],
                         ids=["generic", "lean_bk64", "lean_bk32", "lean_two_samples_per_tile", "lean_two_samples_bk32"])
def test_gemm_nt_grn_prologue(dt, M, N, K, hw):
    H = _hip()
    A, Bw = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, seed=2, scale=K**-0.5)
    s, beta = 1 + 0.3 * rnd(M // hw, K, seed=3), 0.1 * rnd(K, seed=4)
    res, bias = rnd(M, N, dt=dt, seed=5), rnd(N, seed=6)

    def run(ops, dev):
        C = torch.zeros(M, N, dtype=dt, device=dev)
        ops.gemm("nt", A.to(dev), Bw.to(dev), C, M, N, K, K, K, N, dtype=dt, pro=R.PRO_GRN, grn_s=s.to(dev),
                 grn_b=beta.to(dev), hw=hw, epi=R.EPI_BIAS_RES, bias=bias.to(dev), res=res.to(dev), ldr=N)
        return C

    close(run(H, DEV), run(R, "cpu"), dt, "grn-prologue gemm")


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_gemm_nt_per_sample_weights_fold_grn(dt):
    """VsxGemm.b_bstride + vsx_scale_weight_samples: (g·s_b + β)·W2ᵀ as a plain GEMM with per-sample weights
    W2·diag(s_b) and bias b2 + W2·β — against the ref backend AND against the GRN-prologue formulation it replaces."""
    H = _hip()
    M, N, K, hw = 768, 96, 384, 256
    nb = M // hw
    A = rnd(M, K, dt=dt, seed=1)
    W2 = rnd(N, K, seed=2, scale=K**-0.5)            # fp32 master weights
    s, beta = 1 + 0.3 * rnd(nb, K, seed=3), 0.1 * rnd(K, seed=4)
    res, bias = rnd(M, N, dt=dt, seed=5), rnd(N, seed=6)

    def run(ops, dev):
        Ws = ops.scale_weight_samples(W2.to(dev), s.to(dev), dt)
        b2 = ops.matvec(W2.to(dev), beta.to(dev), bias.to(dev), N, K)
        C = torch.zeros(M, N, dtype=dt, device=dev)
        ops.gemm("nt", A.to(dev), Ws, C, M, N, K, K, K, N, dtype=dt, hw=hw, b_bstride=N * K, epi=R.EPI_BIAS_RES, bias=b2,
                 res=res.to(dev), ldr=N)
        return Ws, C

    (Wg, Cg), (Wr, Cr) = run(H, DEV), run(R, "cpu")
    assert Wg.shape == (nb, N, K) and Wg.dtype == dt
    close(Wg, Wr, dt, "scaled weights")
    close(Cg, Cr, dt, "folded fc2")
    C2 = torch.zeros(M, N, dtype=dt, device=DEV)
    H.gemm("nt", A.to(DEV), W2.to(dt).to(DEV), C2, M, N, K, K, K, N, dtype=dt, pro=R.PRO_GRN, grn_s=s.to(DEV), grn_b=beta.to(DEV),
           hw=hw, epi=R.EPI_BIAS_RES, bias=bias.to(DEV), res=res.to(DEV), ldr=N)
    close(Cg, C2, dt, "fold == prologue")
    with pytest.raises(RuntimeError, match="per-sample weights"):
        H.gemm("nt", A.to(DEV), Wg, C2, M, 40, K, K, K, 40, dtype=dt, hw=hw, b_bstride=N * K)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("B,gh,gw,cin,cout", [(2, 6, 5, 16, 24),      # generic kernel (cs % 32 != 0)
                                              (3, 8, 8, 96, 192),     # lean kernel (round 6), 32-deep slabs, tiles across samples
                                              (2, 16, 12, 64, 96),    # lean, 64-deep slabs, ragged last M tile (384 rows)
                                              (5, 4, 4, 32, 80),      # lean, 16-pixel samples: 8 samples per 128-row tile
                                              (1, 20, 36, 192, 384)]) # lean, wide grid (a tile spans < 4 grid rows)
def test_gemm_nt_patch2_gather_and_scatter(dt, B, gh, gw, cin, cout):
    H = _hip()
    M = B * gh * gw
    src = rnd(B * 4 * gh * gw, cin, dt=dt, seed=1)
    Wd = rnd(cout, 4 * cin, dt=dt, seed=2, scale=0.2)
    bias = rnd(cout, seed=3)
    d = rnd(M, cout, dt=dt, seed=4)
    WT = Wd.t().contiguous()

    def run(ops, dev):
        C = torch.zeros(M, cout, dtype=dt, device=dev)
        ops.gemm("nt", src.to(dev), Wd.to(dev), C, M, cout, 4 * cin, cin, 4 * cin, cout, dtype=dt, a_mode=R.A_PATCH2,
                 gh=gh, gw=gw, cs=cin, epi=R.EPI_BIAS, bias=bias.to(dev))
        dx = torch.zeros(B * 4 * gh * gw, cin, dtype=dt, device=dev)
        ops.gemm("nt", d.to(dev), WT.to(dev), dx, M, 4 * cin, cout, cout, cout, cin, dtype=dt, c_mode=R.A_PATCH2,
                 c_cs=cin, gh=gh, gw=gw)
        return C, dx

    (Cg, dxg), (Cr, dxr) = run(H, DEV), run(R, "cpu")
    close(Cg, Cr, dt, "patch2 gather")
    close(dxg, dxr, dt, "patch2 scatter")


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_gemm_conv3_z_batched(dt):
    """the head 3x3x3 convolution as 5 z-batched implicit GEMMs + its data / weight gradients"""
    H = _hip()
    B, gh, gw, c3, cmid, Zo = 2, 9, 7, 8, 32, 5
    D7 = Zo + 2
    Mh = B * gh * gw
    hin = rnd(Mh, D7 * c3, dt=dt, seed=1)
    Wc = rnd(cmid, c3, 3, 3, 3, seed=2, scale=0.1)
    bias = rnd(cmid, seed=3)
    dU = rnd(Mh, Zo * cmid, dt=dt, seed=4)
    zs = [min(max(zp - 2, 0), Zo - 3) for zp in range(D7)]

    def run(ops, dev):
        Wg, _ = ops.prep_weight(Wc.to(dev), cmid, c3, 27, dt, tapmode=1)
        Wdg = ops.prep_head_dgrad(Wc.to(dev), cmid, c3, Zo, dt)
        U = torch.zeros(Mh, Zo * cmid, dtype=dt, device=dev)
        st = torch.zeros(2, B, cmid, device=dev)
        ops.gemm_z("nt", hin.to(dev), Wg, U, Mh, cmid, 27 * c3, D7 * c3, 27 * c3, Zo * cmid, dtype=dt, a_mode=R.A_CONV3,
                   gh=gh, gw=gw, cs=3 * c3, nz=Zo, a_coff=[z * c3 for z in range(Zo)], b_off=[0] * Zo,
                   c_coff=[z * cmid for z in range(Zo)], epi=R.EPI_BIAS_STATS, bias=bias.to(dev), red0=st[0], red1=st[1],
                   hw=gh * gw)
        dWc = torch.zeros(cmid, 27 * c3, device=dev)
        db = torch.zeros(cmid, device=dev)
        ops.gemm_z("tn", hin.to(dev), dU.to(dev), dWc, Mh, cmid, 27 * c3, D7 * c3, Zo * cmid, 27 * c3, dtype=dt,
                   a_mode=R.A_CONV3, gh=gh, gw=gw, cs=3 * c3, nz=Zo, a_coff=[z * c3 for z in range(Zo)],
                   b_off=[z * cmid for z in range(Zo)], c_coff=[0] * Zo, colsum=db)
        dWp = torch.zeros(cmid, c3, 3, 3, 3, device=dev)
        ops.unprep_grad(dWc, dWp, cmid, c3, 27, tapmode=1)
        dhin = torch.zeros(Mh, D7 * c3, dtype=dt, device=dev)
        ops.gemm_z("nt", dU.to(dev), Wdg, dhin, Mh, c3, 27 * cmid, Zo * cmid, 27 * cmid, D7 * c3, dtype=dt,
                   a_mode=R.A_CONV3, gh=gh, gw=gw, cs=3 * cmid, nz=D7, a_coff=[z * cmid for z in zs],
                   b_off=[zp * c3 * 27 * cmid for zp in range(D7)], c_coff=[zp * c3 for zp in range(D7)])
        return U, st, dWp, db, dhin

    g, r = run(H, DEV), run(R, "cpu")
    for name, a, b in zip(["U", "stats", "dW", "db", "dhin"], g, r):
        close(a, b, dt, name)
    # independent check of the whole construction against F.conv3d autograd (fp32 only)
    if dt == torch.float32:
        x5 = hin.view(B, gh, gw, D7, c3).permute(0, 4, 3, 1, 2).clone().requires_grad_(True)  # B, c3, D7, H, W
        w = Wc.clone().requires_grad_(True)
        bb = bias.clone().requires_grad_(True)
        y = torch.nn.functional.conv3d(x5, w, bb, padding=(0, 1, 1))  # B, cmid, Zo, H, W
        Uref = y.permute(0, 3, 4, 2, 1).reshape(Mh, Zo * cmid)
        close(g[0], Uref, dt, "U vs conv3d")
        y.backward(dU.view(B, gh, gw, Zo, cmid).permute(0, 4, 3, 1, 2))
        close(g[2], w.grad, dt, "dW vs conv3d")
        close(g[4], x5.grad.permute(0, 3, 4, 2, 1).reshape(Mh, D7 * c3), dt, "dhin vs conv3d")


@pytest.mark.parametrize("B,gh,gw", [(2, 32, 48), (1, 16, 16)])
def test_head_conv_direct_bf16(B, gh, gw):
    """csrc/headconv.hip (LDS-tiled direct 3x3x3 conv, bf16 8 -> 32 channels, 5 planes) against F.conv3d autograd on
    the same bf16-rounded operands, and against the z-batched implicit-GEMM path it replaces."""
    H = _hip()
    dt = torch.bfloat16
    c3, cmid, Zo = 8, 32, 5
    D7 = Zo + 2
    Mh = B * gh * gw
    assert H.head_conv_supported(gh, gw, c3, cmid, Zo, dt) and not H.head_conv_supported(gh + 8, gw, c3, cmid, Zo, dt)
    assert not H.head_conv_supported(gh, gw, c3, cmid, Zo, torch.float32)
    hin = rnd(Mh, D7 * c3, dt=dt, seed=1)
    Wc = rnd(cmid, c3, 3, 3, 3, seed=2, scale=0.1)
    bias = rnd(cmid, seed=3)
    dU = rnd(Mh, Zo * cmid, dt=dt, seed=4)
    Wg, _ = H.prep_weight(Wc.to(DEV), cmid, c3, 27, dt, tapmode=1)
    st = torch.zeros(2, B, cmid, device=DEV)
    U = H.head_conv_fwd(hin.to(DEV), Wg, bias.to(DEV), st[0], st[1], B, gh, gw, c3, cmid, Zo)
    dWc = torch.zeros(cmid, 27 * c3, device=DEV)
    db = torch.zeros(cmid, device=DEV)
    H.head_conv_wgrad(hin.to(DEV), dU.to(DEV), dWc, db, B, gh, gw, c3, cmid, Zo)
    dWp = torch.zeros(cmid, c3, 3, 3, 3, device=DEV)
    H.unprep_grad(dWc, dWp, cmid, c3, 27, tapmode=1)
    dhin = H.head_conv_dgrad(dU.to(DEV), H.head_conv_dgrad_prep(Wg), B, gh, gw, c3, cmid, Zo)
    assert U.dtype == dt and dhin.dtype == dt and dhin.shape == (Mh, D7 * c3)

    # reference: fp32 conv3d on the bf16-rounded operands (what both device paths see)
    x5 = hin.float().view(B, gh, gw, D7, c3).permute(0, 4, 3, 1, 2).clone().requires_grad_(True)
    w = Wc.to(dt).float().clone().requires_grad_(True)
    bb = bias.clone().requires_grad_(True)
    y = torch.nn.functional.conv3d(x5, w, bb, padding=(0, 1, 1))
    Uref = y.permute(0, 3, 4, 2, 1).reshape(Mh, Zo * cmid)
    y.backward(dU.float().view(B, gh, gw, Zo, cmid).permute(0, 4, 3, 1, 2))
    close(U, Uref, dt, "U vs conv3d")
    ur = Uref.detach().to(dt).float().view(B, gh * gw * Zo, cmid)
    close(st[0], ur.sum(1), dt, "sum")
    close(st[1], (ur * ur).sum(1), dt, "sumsq")
    close(dWp, w.grad, dt, "dW vs conv3d")
    close(db, bb.grad, dt, "db vs conv3d")
    close(dhin, x5.grad.permute(0, 3, 4, 2, 1).reshape(Mh, D7 * c3), dt, "dhin vs conv3d")

    # the generic path on the same inputs (same rounding points: fp32 accumulation of bf16 products, one final rounding)
    U2 = torch.zeros(Mh, Zo * cmid, dtype=dt, device=DEV)
    st2 = torch.zeros(2, B, cmid, device=DEV)
    H.gemm_z("nt", hin.to(DEV), Wg, U2, Mh, cmid, 27 * c3, D7 * c3, 27 * c3, Zo * cmid, dtype=dt, a_mode=R.A_CONV3, gh=gh,
             gw=gw, cs=3 * c3, nz=Zo, a_coff=[z * c3 for z in range(Zo)], b_off=[0] * Zo,
             c_coff=[z * cmid for z in range(Zo)], epi=R.EPI_BIAS_STATS, bias=bias.to(DEV), red0=st2[0], red1=st2[1], hw=gh * gw)
    assert (U.float() - U2.float()).abs().max().item() <= 2.0 ** -7 * U2.float().abs().max().item()  # <= 1 bf16 ulp (sum order)
    torch.testing.assert_close(st, st2, rtol=2e-3, atol=2e-2)


# ------------------------------------------------------------------ GEMM tn
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("tr", [1, 0], ids=["tr_read", "scalar_read"])
@pytest.mark.parametrize("M,N,K", [(512, 160, 40), (1000, 96, 384), (70, 8, 32), (9000, 384, 96), (4100, 768, 3072),
                                   (4096, 384, 128), (8192, 224, 896), (2048, 96, 384),  # lean (M % 64 == 0): 64-row steps
                                   (8192, 896, 224), (4096, 256, 520), (4096, 520, 232),  # rectangular 256x128 / 128x256 tiles
                                   (4096, 512, 384), (8192, 768, 192)])  # 256x128 tiles, several of them along N
def test_gemm_tn(dt, tr, M, N, K):
    from viscy_amd import _lib

    H = _hip()
    if SELF_CHECK and M * N * K > 1e9:
        pytest.skip("self-check")
    if K * N > 1_000_000 and dt == torch.float32 and tr == 0:
        pytest.skip("covered by the tr_read id")
    X, Y = rnd(M, N, dt=dt, seed=1), rnd(M, K, dt=dt, seed=2)
    if not SELF_CHECK:
        _lib.lib().vsx_set_flag(b"tn_tr", tr)
    try:
        def run(ops, dev):
            Wt = torch.zeros(N, K, device=dev)
            cs = torch.zeros(N, device=dev)
            ops.gemm("tn", Y.to(dev), X.to(dev), Wt, M, N, K, K, N, K, dtype=dt, colsum=cs)
            return Wt, cs

        (Wg, cg), (Wr, cr) = run(H, DEV), run(R, "cpu")
    finally:
        if not SELF_CHECK:
            _lib.lib().vsx_set_flag(b"tn_tr", 1)
    close(Wg, Wr, dt, "W")
    close(cg, cr, dt, "colsum")


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(16384, 384, 768), (32768, 1536, 384), (16384, 224, 896), (8192, 768, 3072)],
                         ids=["square_tiles", "rect_n_divisible", "rect_n_full", "many_tiles"])
def test_gemm_tn_any_split_count(M, N, K):
    """Round 5: the TN split count is chosen to fill rounds of workgroups (csrc/gemm.hip fill_splits) and is no multiple of 8 any
    more; the XCD-aware workgroup order (tn_xcd_order) must stay a bijection of (tile, split) for every count — a tile computed
    twice or not at all shows up as a wrong weight gradient.  Sweep the split target over odd values, with and without the fill."""
    if SELF_CHECK:
        pytest.skip("HIP-only launch geometry")
    from viscy_amd import _lib, ops

    l, dt = _lib.lib(), torch.bfloat16
    X, Y = rnd(M, N, dt=dt, seed=1).cuda(), rnd(M, K, dt=dt, seed=2).cuda()
    ref = X.double().t() @ Y.double()
    csr = X.double().sum(0)
    saved = {n: l.vsx_get_flag(n) for n in (b"tn_want", b"tn_want2", b"tn_fill")}
    try:
        for fill in (1, 0):
            for want in (97, 200, 333, 555, 768, 1100):
                l.vsx_set_flag(b"tn_fill", fill)
                l.vsx_set_flag(b"tn_want", want)
                l.vsx_set_flag(b"tn_want2", want)
                W, cs = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
                ops.gemm("tn", Y, X, W, M, N, K, K, N, K, dtype=dt, colsum=cs)
                close(W, ref.float(), dt, f"W (tn_fill={fill}, tn_want={want})", scale=ref.abs().max().item())
                close(cs, csr.float(), dt, f"colsum (tn_fill={fill}, tn_want={want})", scale=csr.abs().max().item() + float(M) ** 0.5)
    finally:
        for n, v in saved.items():
            l.vsx_set_flag(n, v)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("M,N,K,hw", [(300, 40, 160, 100), (4096, 224, 896, 1024), (1024, 96, 384, 256), (4096, 896, 224, 1024)],
                         ids=["generic", "lean_wide_rect_n", "lean", "rect_k"])
def test_gemm_tn_grn_and_patch2(dt, M, N, K, hw):
    H = _hip()
    X, Hh = rnd(M, N, dt=dt, seed=1), rnd(M, K, dt=dt, seed=2)
    s, beta = 1 + 0.3 * rnd(M // hw, K, seed=3), 0.1 * rnd(K, seed=4)
    B, gh, gw, cin, cout = 2, 6, 5, 16, 24
    Mp = B * gh * gw
    src, d = rnd(B * 4 * gh * gw, cin, dt=dt, seed=5), rnd(Mp, cout, dt=dt, seed=6)

    def run(ops, dev):
        W1 = torch.zeros(N, K, device=dev)
        ops.gemm("tn", Hh.to(dev), X.to(dev), W1, M, N, K, K, N, K, dtype=dt, pro=R.PRO_GRN, grn_s=s.to(dev),
                 grn_b=beta.to(dev), hw=hw)
        W2 = torch.zeros(cout, 4 * cin, device=dev)
        cs = torch.zeros(cout, device=dev)
        ops.gemm("tn", src.to(dev), d.to(dev), W2, Mp, cout, 4 * cin, cin, cout, 4 * cin, dtype=dt, a_mode=R.A_PATCH2,
                 gh=gh, gw=gw, cs=cin, colsum=cs)
        return W1, W2, cs

    for name, a, b in zip(["grn wgrad", "patch2 wgrad", "colsum"], run(H, DEV), run(R, "cpu")):
        close(a, b, dt, name)


@pytest.mark.parametrize("M,N,K", [(8192, 896, 224), (4096, 768, 192), (16384, 1536, 384), (4096, 640, 200)])
def test_gemm_tn_eight_wave_tiles(M, N, K):
    """tn_rect bits 4 / 5 (round 6, off by default: +-0 on the step): 256 x 256 and 256 x 192 output tiles on eight-wave workgroups
    give the products of the shipped tiles (other fp32 summation order)"""
    from viscy_amd._lib import lib

    H = _hip()
    dt = torch.bfloat16
    X, Y = rnd(M, N, dt=dt, seed=1).to(DEV), rnd(M, K, dt=dt, seed=2).to(DEV)
    old = lib().vsx_get_flag(b"tn_rect")
    res = []
    try:
        for f in (old & ~48, old | 48):
            assert lib().vsx_set_flag(b"tn_rect", f) == 0
            out, cs = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
            H.gemm("tn", Y, X, out, M, N, K, K, N, K, dtype=dt, colsum=cs)
            res.append((out.cpu(), cs.cpu()))
    finally:
        lib().vsx_set_flag(b"tn_rect", old)
    ref = X.float().t().cpu() @ Y.float().cpu()
    for (o, c) in res:
        assert ((o - ref).abs().max() / ref.abs().max()).item() < 2e-5
        assert ((c - X.float().sum(0).cpu()).abs().max() / X.float().sum(0).abs().max().cpu()).item() < 1e-4


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("B,gh,gw,cin,cout", [(4, 8, 8, 96, 192),      # lean TN kernel (round 6): 64-row steps = 8 grid rows, across samples
                                              (3, 32, 32, 32, 96),     # gw = 32: two grid rows per step
                                              (2, 16, 16, 192, 384),   # the 192 -> 384 projection's shape per sample
                                              (6, 4, 4, 64, 128),      # 16-pixel samples
                                              (1, 8, 64, 32, 96),      # gw = 64: one grid row per 64-row step
                                              (2, 4, 128, 64, 128),    # wide grids (the 2048 x 2048 gate shape): a step inside one grid row
                                              (1, 3, 256, 32, 96),
                                              (2, 12, 24, 32, 96)])    # gw divides neither: generic kernel
def test_gemm_tn_patch2_gather_shapes(dt, B, gh, gw, cin, cout):
    """weight gradient of a 2 x 2 stride-2 projection: W[cout, (ky, kx, c)] = sum over output pixels of d^T . patch(src)"""
    H = _hip()
    Mp = B * gh * gw
    src, d = rnd(B * 4 * gh * gw, cin, dt=dt, seed=5), rnd(Mp, cout, dt=dt, seed=6)

    def run(ops, dev):
        W2 = torch.zeros(cout, 4 * cin, device=dev)
        cs = torch.zeros(cout, device=dev)
        ops.gemm("tn", src.to(dev), d.to(dev), W2, Mp, cout, 4 * cin, cin, cout, 4 * cin, dtype=dt, a_mode=R.A_PATCH2,
                 gh=gh, gw=gw, cs=cin, colsum=cs)
        return W2, cs

    for name, a, b in zip(["patch2 wgrad", "colsum"], run(H, DEV), run(R, "cpu")):
        close(a, b, dt, name, scale=b.abs().max().item())


# ------------------------------------------------------------------ LayerNorm / GRN
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("rows,C", [(300, 96), (64, 768), (17, 40), (100, 144), (33, 1024), (50, 224)])
@pytest.mark.parametrize("affine", [True, False])
def test_layernorm_fwd_bwd(dt, rows, C, affine):
    H = _hip()
    x = rnd(rows, C, dt=dt, seed=1, scale=2.0) + 0.5
    dy = rnd(rows, C, dt=dt, seed=2)
    add = rnd(rows, C, dt=dt, seed=3)
    gamma = (1 + 0.2 * rnd(C, seed=4)) if affine else None
    beta = 0.1 * rnd(C, seed=5) if affine else None

    def run(ops, dev):
        mv = lambda t: t.to(dev) if t is not None else None  # noqa: E731
        y, mean, rstd = ops.ln_fwd(mv(x), mv(gamma), mv(beta), rows, C)
        dg = torch.zeros(C, device=dev) if affine else None
        db = torch.zeros(C, device=dev) if affine else None
        if affine:
            dx = ops.ln_bwd(mv(dy), mv(x), mean, rstd, mv(gamma), mv(add), dg, db, rows, C)
        else:  # block LN: backward from the saved normalised activations
            dx = ops.ln_bwd(mv(dy), y, None, rstd, None, None, None, None, rows, C)
        return y, mean, rstd, dx, dg, db

    g, r = run(H, DEV), run(R, "cpu")
    for name, a, b in zip(["y", "mean", "rstd", "dx", "dgamma", "dbeta"], g, r):
        if a is not None:
            close(a, b, dt, name)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_grn_stats_and_gelu_bwd(dt):
    H = _hip()
    B, N, hw = 3, 160, 50
    M = B * hw
    colsq = rnd(B, N, seed=1).abs() * 40 + 1
    colsq[1, 5] = 0.0  # zero-norm channel: gradient must be 0, not NaN
    P = rnd(B, N, seed=2)
    gamma = 0.3 * rnd(N, seed=3)
    dz, h = rnd(M, N, dt=dt, seed=4), rnd(M, N, dt=dt, seed=5, scale=1.5)

    def run(ops, dev):
        s = ops.grn_scale(colsq.to(dev), gamma.to(dev))
        dg, dbt = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        t = ops.grn_bwd_stats(colsq.to(dev), P.to(dev), gamma.to(dev), dg, Sb=P.to(dev) * 0.5, dbeta=dbt)
        d = dz.clone().to(dev)
        cs = torch.zeros(N, device=dev)
        ops.grn_gelu_bwd(d, h.to(dev), s, t, cs, M, N, hw)
        return s, t, dg, d, cs, dbt

    for name, a, b in zip(["s", "t", "dgamma", "dh", "colsum", "dbeta"], run(H, DEV), run(R, "cpu")):
        assert torch.isfinite(a).all(), name
        close(a, b, dt, name)


# ------------------------------------------------------------------ depthwise conv
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("B,Hh,Ww,C", [(2, 16, 16, 96), (1, 8, 12, 40), (2, 4, 4, 192), (1, 2, 2, 768), (1, 33, 19, 24)])
def test_dwconv7(dt, B, Hh, Ww, C):
    H = _hip()
    M = B * Hh * Ww
    x, dy, add = rnd(M, C, dt=dt, seed=1), rnd(M, C, dt=dt, seed=2), rnd(M, C, dt=dt, seed=3)
    w, bias = rnd(49, C, seed=4, scale=0.2), rnd(C, seed=5)

    def run(ops, dev):
        y = ops.dwconv7_fwd(x.to(dev), w.to(dev), bias.to(dev), B, Hh, Ww, C)
        dx = ops.dwconv7_bwd_data(dy.to(dev), w.to(dev), add.to(dev), B, Hh, Ww, C)
        dw, db = torch.zeros(49, C, device=dev), torch.zeros(C, device=dev)
        ops.dwconv7_bwd_weight(dy.to(dev), x.to(dev), dw, db, B, Hh, Ww, C)
        return y, dx, dw, db

    for name, a, b in zip(["y", "dx", "dw", "db"], run(H, DEV), run(R, "cpu")):
        close(a, b, dt, name)


@pytest.mark.parametrize("B,Hh,Ww,C", [(2, 64, 64, 96), (3, 32, 32, 192), (2, 16, 16, 384), (1, 40, 72, 224), (1, 33, 19, 24),
                                       (2, 16, 24, 40), (5, 48, 16, 64)])
def test_dwconv7_matrix_core_path(B, Hh, Ww, C):
    """bf16 forward / data gradient on the matrix cores (banded Toeplitz tiles, dwconv_mfma.hip; flag dw_mfma) against the
    VALU stencil (flag off) and the fp32 reference: whole / partial tiles, one and two x tiles, partial channel slabs,
    several tiles per persistent workgroup.  The MFMA operand is the bf16-rounded weight (the reference's autocast
    precision), so the tight comparison uses bf16-representable weights, where both kernels compute the same products."""
    from viscy_amd._lib import lib

    def set_flag(v):
        assert lib().vsx_set_flag(b"dw_mfma", v) == 0

    H = _hip()
    dt = torch.bfloat16
    M = B * Hh * Ww
    x, dy, add = rnd(M, C, dt=dt, seed=1), rnd(M, C, dt=dt, seed=2), rnd(M, C, dt=dt, seed=3)
    bias = rnd(C, seed=5)
    shipped = 15  # forward / data gradient (bit 0) with LDS-DMA tile fetches (bit 3), weight gradient (bits 1, 2)
    assert lib().vsx_get_flag(b"dw_mfma") == shipped
    for wkind in ("bf16_exact", "fp32"):
        w = rnd(49, C, seed=4, scale=0.2)
        if wkind == "bf16_exact":
            w = w.to(torch.bfloat16).float()
        outs = {}
        for flag in (31, 63, 7, 0):  # LDS-DMA tile fetches wherever they can run (whole 32-channel slabs), the same with two pixels per LDS access (bit 5, round 6), register-staged tiles, VALU stencil
            set_flag(flag)
            try:
                outs[15 if flag == 31 else flag] = (H.dwconv7_fwd(x.to(DEV), w.to(DEV), bias.to(DEV), B, Hh, Ww, C).float().cpu(),
                              H.dwconv7_fwd(x.to(DEV), w.to(DEV), None, B, Hh, Ww, C).float().cpu(),
                              H.dwconv7_bwd_data(dy.to(DEV), w.to(DEV), add.to(DEV), B, Hh, Ww, C).float().cpu(),
                              H.dwconv7_bwd_data(dy.to(DEV), w.to(DEV), None, B, Hh, Ww, C).float().cpu())
            finally:
                set_flag(shipped)
        ref = (R.dwconv7_fwd(x, w, bias, B, Hh, Ww, C).float(), R.dwconv7_fwd(x, w, None, B, Hh, Ww, C).float(),
               R.dwconv7_bwd_data(dy, w, add, B, Hh, Ww, C).float(), R.dwconv7_bwd_data(dy, w, None, B, Hh, Ww, C).float())
        for mm in (15, 7):
            for name, a, b, r in zip(["y", "y_nobias", "dx_add", "dx"], outs[mm], outs[0], ref):
                assert torch.isfinite(a).all(), (mm, name)
                scale = r.abs().max().item()
                close(a, r, dt, f"{name} vs reference ({wkind}, dw_mfma = {mm})")
                if wkind == "bf16_exact":
                    # same products, fp32 accumulation in a different order, one bf16 rounding (two with `add`): a few ulps apart
                    d = (a - b).abs()
                    assert d.max().item() <= (1.6e-2 if name == "dx_add" else 8e-3) * scale, (mm, name, d.max().item() / scale)
                    assert (d > 0).float().mean().item() < (0.5 if name == "dx_add" else 0.2), (mm, name, (d > 0).float().mean().item())
        # the matrix-core kernels run the same MFMA sequence on the same operands: identical bits
        for other in (7, 63):
            for name, a, b in zip(["y", "y_nobias", "dx_add", "dx"], outs[15], outs[other]):
                assert torch.equal(a, b), (other, name, (a - b).abs().max().item())


    # weight gradient: row contraction on the matrix cores (transpose reads) vs the VALU kernel and the fp32 reference —
    # the products are exact in both, only the fp32 summation order differs
    res = {}
    for flag in (7, 3, 0):  # 16-column tiles (shipped), 32-column tiles where the width allows, VALU kernel
        set_flag(flag | (8 if flag else 0))
        try:
            dw, db = torch.zeros(49, C, device=DEV), torch.zeros(C, device=DEV)
            H.dwconv7_bwd_weight(dy.to(DEV), x.to(DEV), dw, db, B, Hh, Ww, C)
            H.dwconv7_bwd_weight(dy.to(DEV), x.to(DEV), dw, db, B, Hh, Ww, C)  # accumulates
            res[flag] = (dw.cpu() / 2, db.cpu() / 2)
        finally:
            set_flag(shipped)
    dwr, dbr = torch.zeros(49, C), torch.zeros(C)
    R.dwconv7_bwd_weight(dy, x, dwr, dbr, B, Hh, Ww, C)
    for name, a, a32, b, r in zip(["dw", "db"], res[7], res[3], res[0], (dwr, dbr)):
        scale = r.abs().max().item()
        bar = 2e-4 * scale + 1e-5 * scale * (M ** 0.5)
        for what, other in (("reference", r), ("VALU kernel", b), ("32-column tiles", a32)):
            assert (a - other).abs().max().item() <= bar, (name, what, (a - other).abs().max().item() / scale)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("B,Hh,Ww,C", [(2, 8, 8, 96), (1, 5, 7, 40), (3, 16, 12, 24), (1, 1, 1, 8)])
def test_conv3x3_patch_gather_and_scatter(dt, B, Hh, Ww, C):
    """vsx_im2col3x3 / vsx_col2im3x3 (the decoder pre-convolution as a GEMM): exact data movement, and together with the
    GEMMs a Conv2d(C, Cout, 3, padding=1) forward / data gradient"""
    H = _hip()
    M = B * Hh * Ww
    x = rnd(M, C, dt=dt, seed=1)
    col = H.im2col3x3(x.to(DEV), B, Hh, Ww, C).cpu()
    assert torch.equal(col, R.im2col3x3(x, B, Hh, Ww, C))
    dcol = rnd(M, 9 * C, dt=dt, seed=2)
    close(H.col2im3x3(dcol.to(DEV), B, Hh, Ww, C), R.col2im3x3(dcol, B, Hh, Ww, C), dt, "col2im")
    # against torch's convolution: K order t*C + c with t = 3*ky + kx == prep_weight(conv.weight, Cout, C, 9)
    Cout = 16
    w = rnd(Cout, C, 3, 3, seed=3, scale=0.1)
    Wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C)
    y = col.float() @ Wp.t()
    yr = torch.nn.functional.conv2d(x.float().view(B, Hh, Ww, C).permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(y, yr.reshape(M, Cout), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ data movement
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_stem_im2col_and_normalize_fusion(dt):
    H = _hip()
    for (B, Cin, Z, Hh, Ww) in [(2, 1, 5, 32, 64), (1, 2, 15, 32, 32)]:
        x = rnd(B, Cin, Z, Hh, Ww, seed=1) * 10 + 3
        sub, div = torch.tensor([1.0, 2.0][:B]), torch.tensor([3.0, 0.5][:B])
        close(H.stem_im2col(x.to(DEV), (5, 4, 4), dt), R.stem_im2col(x, (5, 4, 4), dt), dt, "im2col")
        # rows padded to a whole number of 32-deep MFMA slabs (K = 80 -> 96): zero tail, same head; matching weight helper
        Pr = R.stem_im2col(x, (5, 4, 4), dt)
        KT = Pr.shape[1]
        ldp = (KT + 31) // 32 * 32 + (32 if KT % 32 == 0 else 0)
        Pp = H.stem_im2col(x.to(DEV), (5, 4, 4), dt, ld=ldp)
        assert Pp.shape[1] == ldp > KT and float(Pp[:, KT:].abs().max()) == 0.0
        close(Pp[:, :KT], Pr, dt, "im2col padded")
        wsrc = rnd(24, 80, dt=dt, seed=9)
        wp = H.pad_cols(wsrc.to(DEV), 96)
        assert torch.equal(wp[:, :80].cpu(), wsrc) and float(wp[:, 80:].abs().max()) == 0.0
        close(H.stem_im2col(x.to(DEV), (5, 4, 4), dt, sub.to(DEV), div.to(DEV)), R.stem_im2col(x, (5, 4, 4), dt, sub, div), dt,
              "im2col+normalize")


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("c,cs", [(48, 96), (20, 40), (96, 0), (12, 52), (10, 54), (224, 96)])  # vector form (c, cs whole vectors) and the element-wise fallback
def test_pixel_shuffle_cat(dt, c, cs):
    H = _hip()
    if (c + cs) % (8 if dt == torch.bfloat16 else 4) and not SELF_CHECK:
        with pytest.raises(RuntimeError, match="multiple of"):  # 16-byte vector contract of the ABI
            H.pixel_shuffle_cat_fwd(rnd(4, 4 * c, dt=dt).to(DEV), rnd(16, cs, dt=dt).to(DEV), 1, 2, 2, c, cs)
        return
    B, h, w = 2, 5, 3
    low = rnd(B * h * w, 4 * c, dt=dt, seed=1)
    skip = rnd(B * 4 * h * w, cs, dt=dt, seed=2) if cs else None
    dcat = rnd(B * 4 * h * w, c + cs, dt=dt, seed=3)
    cat_g = H.pixel_shuffle_cat_fwd(low.to(DEV), skip.to(DEV) if cs else None, B, h, w, c, cs)
    assert torch.equal(cat_g.cpu(), R.pixel_shuffle_cat_fwd(low, skip, B, h, w, c, cs))
    dl_g, ds_g = H.pixel_shuffle_cat_bwd(dcat.to(DEV), B, h, w, c, cs)
    dl_r, ds_r = R.pixel_shuffle_cat_bwd(dcat, B, h, w, c, cs)
    assert torch.equal(dl_g.cpu(), dl_r)
    if cs:
        assert torch.equal(ds_g.cpu(), ds_r)


@pytest.mark.parametrize("geom", [(2, 4, 6), (2, 5, 64), (1, 20, 128)], ids=["tiles", "strip", "strips-3-row-ranges"])
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("pool", [True, False])
def test_head_shuffle(dt, pool, geom):
    """`strip*`: 64 | w, so the pooled bf16 permutations run on column strips (csrc/spatial.hip, `head_rows` bits 3 / 4); the
    20-row case splits every strip over three workgroups (the row above a range only feeds the carried sums)."""
    H = _hip()
    (B, h, w), C3, D = geom, 8, 7
    dec = rnd(B * h * w, 4 * C3 * D, dt=dt, seed=1)
    dh = rnd(B * 4 * h * w, C3 * D, dt=dt, seed=2)
    close(H.head_shuffle_fwd(dec.to(DEV), B, h, w, C3, D, pool), R.head_shuffle_fwd(dec, B, h, w, C3, D, pool), dt, "fwd")
    close(H.head_shuffle_bwd(dh.to(DEV), B, h, w, C3, D, pool), R.head_shuffle_bwd(dh, B, h, w, C3, D, pool), dt, "bwd")


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("pool", [True, False])
def test_voxel_shuffle_head(dt, pool):
    """PixelToVoxelShuffleHead (FCMAE): pixel shuffle x4 + MONAI pad-pool + reshape, forward and its transpose."""
    H = _hip()
    B, h, w, Cout, D, s = 2, 5, 7, 2, 3, 4
    feat = rnd(B * h * w, Cout * D * s * s, dt=dt, seed=1)
    dout = rnd(B, Cout, D, s * h, s * w, seed=2)
    og = H.voxel_shuffle_fwd(feat.to(DEV), B, h, w, Cout, D, s, pool)
    orf = R.voxel_shuffle_fwd(feat, B, h, w, Cout, D, s, pool)
    assert og.shape == (B, Cout, D, s * h, s * w) and og.dtype == torch.float32
    close(og, orf, dt, "voxel shuffle fwd")
    close(H.voxel_shuffle_bwd(dout.to(DEV), B, h, w, Cout, D, s, pool, dt), R.voxel_shuffle_bwd(dout, B, h, w, Cout, D, s, pool, dt),
          dt, "voxel shuffle bwd")


@pytest.mark.parametrize("geom", [(2, 6, 10, 5, 32, 2), (2, 3, 128, 5, 32, 2), (1, 4, 64, 3, 64, 4)],
                         ids=["thread-per-voxel", "row-tiles", "row-tiles-mid64"])
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_head_tail_fwd_bwd(dt, geom):
    """`row-tiles*`: W2 is a multiple of 64, so the bf16 forward and backward pass 2 run the row-tiled MFMA kernels
    (csrc/head.hip, `head_rows` bits 0 / 1); fp32 stays on the thread-per-voxel kernels at every geometry."""
    H = _hip()
    B, H2, W2, Z, Cmid, Cout = geom
    Mh = B * H2 * W2
    U = rnd(Mh, Z * Cmid, dt=dt, seed=1, scale=2.0) + 0.3
    w2, b2, alpha = rnd(4 * Cout, Cmid, seed=2, scale=0.2), rnd(4 * Cout, seed=3), torch.tensor([0.25])
    dout = rnd(B, Cout, Z, 2 * H2, 2 * W2, seed=4)
    u3 = U.float().view(B, H2 * W2 * Z, Cmid)
    # PReLU has a kink at n̂ = 0: keep every normalised value clear of it so that last-bit differences in the
    # statistics cannot flip a derivative branch (the comparison would be ill-conditioned, not wrong)
    nh = (u3 - u3.mean(1, keepdim=True)) / u3.std(1, keepdim=True)
    U = (u3 + 0.05 * (nh.abs() < 5e-3) * u3.std(1, keepdim=True)).view(Mh, Z * Cmid).to(dt)
    u3 = U.float().view(B, H2 * W2 * Z, Cmid)
    ssum, ssq = u3.sum(1), (u3 * u3).sum(1)

    def run(ops, dev):
        mv = lambda t: t.to(dev)  # noqa: E731
        out = ops.head_out_fwd(mv(U), mv(ssum), mv(ssq), mv(w2), mv(b2), mv(alpha), B, H2, W2, Z, Cmid, Cout)
        S = torch.zeros(2, B, Cmid, device=dev)
        dal = torch.zeros(1, device=dev)
        act, dv = ops.head_out_bwd1(mv(U), mv(ssum), mv(ssq), mv(w2), mv(alpha), mv(dout), S[0], S[1], dal, B, H2, W2, Z,
                                    Cmid, Cout)
        dU = ops.head_out_bwd2(mv(U), mv(ssum), mv(ssq), mv(w2), mv(alpha), dv, S[0], S[1], B, H2, W2, Z, Cmid, Cout)
        return out, act, dv, S, dal, dU

    g, r = run(H, DEV), run(R, "cpu")
    for name, a, b in zip(["out", "act", "dv", "S", "dalpha", "dU"], g, r):
        close(a, b, dt, name)
    if dt == torch.float32:  # independent: autograd through InstanceNorm3d + PReLU + 1x1x1 conv + pixel shuffle
        Ur = U.clone().requires_grad_(True)
        al = alpha.clone().requires_grad_(True)
        out_ref, _ = R._head_full(Ur, w2, b2, al, B, H2, W2, Z, Cmid, Cout, 1e-5)
        close(g[0], out_ref, dt, "out vs autograd")
        out_ref.backward(dout)
        close(g[5], Ur.grad, dt, "dU vs autograd")
        close(g[4], al.grad, dt, "dalpha vs autograd")


@pytest.mark.parametrize("cfg", [(2, 6, 10, 5, 32, 2), (3, 9, 7, 3, 32, 2), (1, 40, 52, 5, 64, 4), (2, 64, 80, 5, 32, 2),
                                 (2, 5, 64, 5, 32, 2), (1, 70, 128, 5, 32, 2), (2, 9, 64, 3, 32, 2), (1, 6, 128, 5, 64, 4)],
                         ids=["mid32", "mid32-ragged", "mid64-multi-iter", "mid32-multi-iter", "rows", "rows-multi-tile", "rows-z3",
                              "rows-mid64"])
def test_head_bwd1_with_folded_weight_gradient(cfg):
    """bf16 pass 1 with the 1x1x1 weight gradient folded in (voxel contraction on MFMA inside the kernel) against
    pass 1 + the explicit dv^T . act product; dW2 / db2 are accumulated into (non-zero start)."""
    H = _hip()
    dt = torch.bfloat16
    B, H2, W2, Z, Cmid, Cout = cfg
    Mh = B * H2 * W2
    U = rnd(Mh, Z * Cmid, dt=dt, seed=11, scale=2.0) + 0.3
    w2, alpha = rnd(4 * Cout, Cmid, seed=12, scale=0.2), torch.tensor([0.25])
    dout = rnd(B, Cout, Z, 2 * H2, 2 * W2, seed=14)
    u3 = U.float().view(B, H2 * W2 * Z, Cmid)
    ssum, ssq = u3.sum(1), (u3 * u3).sum(1)
    w0, b0 = rnd(4 * Cout, Cmid, seed=15), rnd(4 * Cout, seed=16)

    def run(ops, dev):
        mv = lambda t: t.to(dev)  # noqa: E731
        S = torch.zeros(2, B, Cmid, device=dev)
        dal = torch.zeros(1, device=dev)
        dW, db = mv(w0).clone(), mv(b0).clone()
        dv = ops.head_out_bwd1_wgrad(mv(U), mv(ssum), mv(ssq), mv(w2), mv(alpha), mv(dout), S[0], S[1], dal, dW, db, B, H2, W2,
                                     Z, Cmid, Cout)
        return dv, S, dal, dW - mv(w0), db - mv(b0)

    g, r = run(H, DEV), run(R, "cpu")
    if W2 % 64 == 0:  # the row-tiled pass again with two workgroups per sample: each walks a run of tiles (the software pipeline)
        from viscy_amd import _lib as L

        assert L.lib().vsx_set_flag(b"head_bps", 2) == 0
        try:
            g2 = run(H, DEV)
        finally:
            L.lib().vsx_set_flag(b"head_bps", 0)
        assert torch.equal(g2[0], g[0]), "dv, several tiles per workgroup"
        for name, a, b in zip(("S", "dalpha", "dW2", "db2"), g2[1:], g[1:]):
            close(a, b, dt, name + ", several tiles per workgroup")
    assert torch.equal(g[0].cpu(), r[0]), "dv"
    close(g[1], r[1], dt, "S")
    close(g[2], r[2], dt, "dalpha")
    # fp32 accumulation of identical bf16 products: only the summation order differs
    for name, a, b in (("dW2", g[3], r[3]), ("db2", g[4], r[4])):
        err = (a.cpu() - b).abs().max().item()
        assert err <= 2e-5 * max(1.0, b.abs().max().item()) * (H2 * W2 * Z * B) ** 0.5, (name, err)
    # and the unfused HIP pair gives the same gradient
    S = torch.zeros(2, B, Cmid, device=DEV)
    act, dv = H.head_out_bwd1(U.to(DEV), ssum.to(DEV), ssq.to(DEV), w2.to(DEV), alpha.to(DEV), dout.to(DEV), S[0], S[1],
                              torch.zeros(1, device=DEV), B, H2, W2, Z, Cmid, Cout)
    ref = dv.float().t() @ act.float()
    assert (g[3] - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------ parameter space
def test_prep_unprep_adamw():
    H = _hip()
    dt = torch.bfloat16
    W = rnd(24, 16, 2, 2, seed=1)
    gamma = 1 + 0.1 * rnd(16, seed=2)
    for ops, dev in ((H, DEV), (R, "cpu")):
        pass
    dg, dgT = H.prep_weight(W.to(DEV), 24, 16, 4, dt, want=True, want_t=True, gamma=gamma.to(DEV))
    rg, rgT = R.prep_weight(W, 24, 16, 4, dt, want=True, want_t=True, gamma=gamma)
    assert torch.equal(dg.cpu(), rg) and torch.equal(dgT.cpu(), rgT)
    g = rnd(24, 64, seed=3)
    u, beta = rnd(40, seed=4), rnd(16, seed=5)
    W1 = rnd(40, 16, seed=6)
    g1 = rnd(40, 16, seed=7)
    outs = []
    for ops, dev in ((H, DEV), (R, "cpu")):
        dp = torch.zeros(24, 16, 2, 2, device=dev)
        ops.unprep_grad(g.to(dev), dp, 24, 16, 4)
        dp1, dgam = torch.zeros(40, 16, device=dev), torch.zeros(16, device=dev)
        ops.unprep_grad(g1.to(dev), dp1, 40, 16, 1, gamma=gamma.to(dev), W=W1.to(dev), dgamma=dgam, u=u.to(dev),
                        beta=beta.to(dev))
        mv = ops.matvec(W1.to(dev), beta.to(dev), u.to(dev), 40, 16)
        acc = torch.zeros(16, device=dev)
        ops.matvec_t_add(W1.to(dev), u.to(dev), acc, 40, 16)
        tr = torch.zeros(49, 24, device=dev)
        ops.transpose_f32(rnd(24, 49, seed=8).to(dev), tr, 24, 49, False)
        outs.append((dp, dp1, dgam, mv, acc, tr))
    for name, a, b in zip(["unprep", "unprep-fold", "dgamma", "matvec", "matvec_t", "transpose"], *outs):
        close(a, b, torch.float32, name)
    # AdamW: 3 steps vs torch.optim.AdamW
    n = 1003
    p0, grads = rnd(n, seed=10), [rnd(n, seed=11 + i) for i in range(3)]
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-2)
    p = p0.clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i, gr in enumerate(grads):
        pt.grad = gr.clone()
        opt.step()
        t = i + 1
        hyper = torch.tensor([1e-2, 0.9, 0.999, 1e-8, 0.01, 1 - 0.9**t, 1 - 0.999**t, 1.0]).to(DEV)
        H.adamw(p, gr.to(DEV), m, v, hyper)
    close(p, pt.detach(), torch.float32, "adamw")


# ---------------------------------------------------------------- FCMAE masked pre-training pieces (csrc/mae.hip)
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("C", [96, 40, 6, 3])
def test_rows_select_gather_scatter_mask(dt, C):
    """masked_patchify / masked_unpatchify / `x *= unmasked` as one row permutation; every vector width (16/8/4/2-byte rows)."""
    H = _hip()
    n, k = 1000, 380
    g = torch.Generator().manual_seed(C)
    src = rnd(n, C, dt=dt, seed=1)
    kept = torch.randperm(n, generator=g)[:k].sort().values.int()
    inv = torch.full((n,), -1, dtype=torch.int32)
    inv[kept.long()] = torch.arange(k, dtype=torch.int32)
    keep = torch.where(inv >= 0, torch.arange(n, dtype=torch.int32), torch.tensor(-1, dtype=torch.int32))
    comp = H.rows_select(src.to(DEV), kept.to(DEV), k, C)
    assert torch.equal(comp.cpu(), src[kept.long()])                                   # bit-exact data movement
    dense = H.rows_select(comp, inv.to(DEV), n, C)
    assert torch.equal(dense.cpu(), R.rows_select(src[kept.long()], inv, n, C))
    assert torch.equal(H.rows_select(src.to(DEV), keep.to(DEV), n, C).cpu(), dense.cpu())
    if (C * src.element_size()) % 4 == 0:
        add = rnd(k, C, dt=dt, seed=2)
        close(H.rows_select(src.to(DEV), kept.to(DEV), k, C, add=add.to(DEV)), R.rows_select(src, kept, k, C, add=add), dt, "gather+add")


@pytest.mark.parametrize("shape", [(2, 1, 5, 64, 96), (3, 2, 3, 32, 36), (1, 2, 1, 256, 256)])
def test_masked_mse_loss_vs_oracle(shape):
    """cytoland MaskedMSELoss: value and gradient vs the oracle restatement (pinned on the reference's own class, G9)."""
    from oracle.fcmae_ref import MaskedMSELoss as RefLoss
    from viscy_amd.losses import MaskedMSELoss

    B, C, Z, Hh, Ww = shape
    g = torch.Generator().manual_seed(7)
    p = torch.randn(shape, generator=g, requires_grad=True)
    o = torch.randn(shape, generator=g)
    m = torch.rand(B, 1, Hh, Ww, generator=g) < 0.4
    lr = RefLoss()(p, o, m)
    lr.backward()
    pc = p.detach().to(DEV).requires_grad_(True)
    lg = MaskedMSELoss()(pc, o.to(DEV), m.to(DEV))
    (3.0 * lg).backward()
    assert abs(lg.item() - lr.item()) <= 2e-6 * abs(lr.item())
    torch.testing.assert_close(pc.grad.cpu(), 3.0 * p.grad, rtol=1e-5, atol=1e-9)
    with pytest.raises(ValueError, match="mask must be"):
        MaskedMSELoss()(pc, o.to(DEV), m[:, :, :-1].to(DEV))


# ---------------------------------------------------------------- DynaCLR tail + loss (csrc/contrastive.hip)
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_avgpool_rows(dt):
    H = _hip()
    B, hw, C = 5, 36, 200
    x, dout = rnd(B * hw, C, dt=dt, seed=1), rnd(B, C, seed=2)
    close(H.avgpool_rows_fwd(x.to(DEV), B, hw, C), R.avgpool_rows_fwd(x, B, hw, C), torch.float32, "avgpool fwd")
    close(H.avgpool_rows_bwd(dout.to(DEV), B, hw, C, dt), R.avgpool_rows_bwd(dout, B, hw, C, dt), dt, "avgpool bwd")


@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
@pytest.mark.parametrize("relu", [True, False], ids=["relu", "plain"])
def test_bn1d_fwd_bwd_vs_torch(training, relu):
    """nn.BatchNorm1d semantics incl. the running-statistics update (momentum 0.1, unbiased variance)"""
    H = _hip()
    B, F = 12, 300
    x, w, b = rnd(B, F, seed=1, scale=2.0) + 0.5, 1 + 0.2 * rnd(F, seed=2), 0.3 * rnd(F, seed=3)
    rm, rv = 0.1 * rnd(F, seed=4), 0.5 + torch.rand(F, generator=torch.Generator().manual_seed(5))
    dy = rnd(B, F, seed=6)
    bn = torch.nn.BatchNorm1d(F)
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train(training)
    xr = x.clone().requires_grad_(True)
    yr = bn(xr)
    if relu:
        yr = yr.relu()
    yr.backward(dy)
    rm_g, rv_g = rm.clone().to(DEV), rv.clone().to(DEV)
    y, sm, sr = H.bn1d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), rm_g, rv_g, training, relu)
    torch.testing.assert_close(y.cpu(), yr.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rm_g.cpu(), bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv_g.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    dw, db = torch.ones(F, device=DEV), torch.ones(F, device=DEV)  # the backward ADDS into the gradient buffers
    dx = H.bn1d_bwd(dy.to(DEV), x.to(DEV), y, w.to(DEV), sm, sr, dw, db, training, relu)
    torch.testing.assert_close(dx.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dw.cpu() - 1, bn.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu() - 1, bn.bias.grad, rtol=1e-4, atol=1e-4)
    if training:
        with pytest.raises(RuntimeError, match="more than 1 value per channel"):
            H.bn1d_fwd(x[:1].to(DEV), w.to(DEV), b.to(DEV), rm_g, rv_g, True, relu)


def test_ntxent_matches_reference_golden():
    """values and gradients stored from the REFERENCE's NTXentHCL (G10), beta = 0 and beta > 0"""
    from viscy_amd.contrastive import NTXentHCL, NTXentLoss

    for c in load_golden("contrastive.pt")["loss"].values():
        e = torch.randn(2 * c["n"], c["dim"], generator=torch.Generator().manual_seed(c["seed"]))
        labels = torch.cat((torch.arange(c["n"]), torch.arange(c["n"])))
        eg = e.to(DEV).requires_grad_(True)
        loss = NTXentHCL(temperature=c["temperature"], beta=c["beta"])(eg, labels.to(DEV))
        assert abs(loss.item() - c["loss"]) <= 2e-5 * abs(c["loss"]), c
        (2.0 * loss).backward()
        torch.testing.assert_close(eg.grad.cpu(), 2.0 * c["grad"], rtol=2e-4, atol=1e-6 + 2e-4 * c["grad"].abs().max().item())
        if c["beta"] == 0.0:
            l0 = NTXentLoss(temperature=c["temperature"])(e.to(DEV), labels.to(DEV))
            assert abs(l0.item() - c["loss"]) <= 2e-5 * abs(c["loss"])


@pytest.mark.parametrize("beta", [0.0, 0.7])
def test_ntxent_general_labels_vs_oracle(beta):
    """several positives per anchor, unequal class sizes, a large batch (N = 600 > one 256-thread sweep)"""
    from oracle.contrastive_ref import NTXentHCL as RefLoss
    from viscy_amd.contrastive import NTXentHCL

    g = torch.Generator().manual_seed(9)
    for N, D, ncls in [(37, 24, 9), (600, 128, 300)]:
        e = torch.randn(N, D, generator=g, requires_grad=True)
        labels = torch.randint(0, ncls, (N,), generator=g)
        lr = RefLoss(temperature=0.2, beta=beta)(e, labels)
        lr.backward()
        eg = e.detach().to(DEV).requires_grad_(True)
        lg = NTXentHCL(temperature=0.2, beta=beta)(eg, labels.to(DEV))
        lg.backward()
        assert abs(lg.item() - lr.item()) <= 5e-5 * abs(lr.item()), (N, lg.item(), lr.item())
        torch.testing.assert_close(eg.grad.cpu(), e.grad, rtol=5e-4, atol=1e-7 + 5e-4 * e.grad.abs().max().item())


def test_layer_scale_fold_unfold():
    H = _hip()
    Rr, K = 37, 148
    W, b, gam = rnd(Rr, K, seed=1), rnd(Rr, seed=2), 0.5 + 0.2 * rnd(Rr, seed=3)
    dWs, dbs = rnd(Rr, K, seed=4), rnd(Rr, seed=5)
    Ws, bs = H.layer_scale_fold(*cuda(W, b, gam))
    Wr, br = R.layer_scale_fold(W, b, gam)
    assert torch.equal(Ws.cpu(), Wr) and torch.equal(bs.cpu(), br)
    outs = []
    for ops, dev in ((R, "cpu"), (H, DEV)):
        dW, db, dg = torch.ones(Rr, K, device=dev), torch.ones(Rr, device=dev), torch.ones(Rr, device=dev)
        ops.layer_scale_unfold(dWs.to(dev), dbs.to(dev), W.to(dev), b.to(dev), gam.to(dev), dW, db, dg)
        outs.append((dW.cpu(), db.cpu(), dg.cpu()))
    for a, c in zip(outs[1], outs[0]):
        torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ fused GRN-MLP (csrc/mlp.hip)
@pytest.mark.parametrize("C,hw,B", [(96, 256, 3), (96, 1024, 2), (192, 256, 2), (224, 512, 2), (384, 256, 3), (384, 256, 2)])
@pytest.mark.parametrize("drop_path", [False, True], ids=["plain", "droppath"])
def test_fused_grn_mlp_matches_unfused_kernels_and_reference(C, hw, B, drop_path):
    """vsx_mlp_fwd (hidden activation on chip, fc1 recomputed in the output pass) against (a) the fp32 statement with the
    same bf16 rounding points (h, g and the GRN output are bf16 values in both schedules) and (b) the unfused HIP kernels"""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    M, H4 = B * hw, 4 * C
    saved = L.lib().vsx_get_flag(b"mlp_fused")
    L.lib().vsx_set_flag(b"mlp_fused", 15)  # bit 2: the C = 384 training passes too (their inference pair was removed in round 4)
    try:
        assert ops.mlp_supported(C, hw, M, dt) == (C != 384) and ops.mlp_supported(C, hw, M, dt, 2)
        _fused_mlp_case(C, hw, B, drop_path, dt, M, H4, L, ops)
    finally:
        L.lib().vsx_set_flag(b"mlp_fused", saved)


def _fused_mlp_case(C, hw, B, drop_path, dt, M, H4, L, ops):
    xh = rnd(M, C, dt=dt, seed=1).cuda()
    res = rnd(M, C, dt=dt, seed=2).cuda()
    W1 = rnd(H4, C, dt=dt, seed=3, scale=C ** -0.5).cuda()
    W2 = rnd(C, H4, dt=dt, seed=4, scale=H4 ** -0.5).cuda()
    b1, b2 = rnd(H4, seed=5, scale=0.1).cuda(), rnd(C, seed=6, scale=0.1).cuda()
    gamma, beta = rnd(H4, seed=7, scale=0.3).cuda(), rnd(H4, seed=8, scale=0.1).cuda()
    rs = (torch.tensor([0.0, 1.25, 1.25][:B] + [1.25] * max(0, B - 3))[:B]).cuda() if drop_path else None
    img = ops.mlp_pack(W1, W2, C)
    colsq = torch.zeros((B, H4), dtype=torch.float32, device="cuda")
    inference_pair = ops.mlp_supported(C, hw, M, dt)   # C <= 224; the C = 384 blocks run the training fc1 (MODE 2) only
    if inference_pair:
        ops.mlp_stats(xh, img, b1, colsq, M, C, hw)
    else:
        ops.mlp_fc1(xh, img, b1, colsq, M, C, hw)
    s = ops.grn_scale(colsq, gamma)
    out = ops.mlp_out(xh, img, b1, s, beta, b2, res, rs, M, C, hw) if inference_pair else None
    # (a) fp32 statement with the kernels' rounding points
    h = (xh.float() @ W1.float().T + b1).to(dt).float()
    g = torch.nn.functional.gelu(h).to(dt).float()
    colsq_ref = (g * g).view(B, hw, H4).sum(1)
    close(colsq, colsq_ref, torch.float32, "colsq", scale=colsq_ref.abs().max().item() * 5)  # bf16 ulp flips of g at the GELU rounding
    s_ref = 1 + gamma * colsq_ref.sqrt() / (colsq_ref.sqrt().mean(1, keepdim=True) + 1e-6)
    z = (g.view(B, hw, H4) * s_ref[:, None] + beta).to(dt).float().view(M, H4)
    branch = z @ W2.float().T + b2
    if rs is not None:
        branch = branch.view(B, hw, C) * rs.view(B, 1, 1)
    ref = (res.float() + branch.reshape(M, C)).to(dt)
    if inference_pair:
        close(out, ref, dt, "fused mlp vs fp32 statement")
    # (b) the unfused HIP schedule
    hh, gg = torch.empty((M, H4), dtype=dt, device="cuda"), torch.empty((M, H4), dtype=dt, device="cuda")
    csq2 = torch.zeros_like(colsq)
    ops.gemm("nt", xh, W1, hh, M, H4, C, C, C, H4, dtype=dt, epi=L.EPI_BIAS_GELU_SQ, bias=b1, red0=csq2, hw=hw, C2=gg)
    assert torch.equal(gg.float(), g.to(dt).float()) or (gg.float() - g).abs().max().item() <= 0.02 * g.abs().max().item()
    close(colsq, csq2, torch.float32, "colsq vs unfused", scale=csq2.abs().max().item() * 5)
    # training fc1 on the fused kernel (MODE 2): the same h / g / colsq as the GEMM with the GELU epilogue (the table GELU is
    # the correctly rounded erf GELU, the epilogue's Abramowitz-Stegun evaluation may differ from it by one bf16 ulp)
    csq3 = torch.zeros_like(colsq)
    h3, g3 = ops.mlp_fc1(xh, img, b1, csq3, M, C, hw)
    assert (h3.float() - hh.float()).abs().max().item() <= 0.008 * hh.float().abs().max().item()   # accumulation order: 1 ulp
    assert ((h3 != hh).float().mean().item()) < 0.02
    same_h = h3 == hh
    dg = (g3.float() - gg.float()).abs()
    assert dg[same_h].max().item() <= 0.008 * gg.float().abs().max().item() and (dg[same_h] > 0).float().mean().item() < 0.01
    gref = torch.nn.functional.gelu(h3.float().double()).float().to(dt).float()  # erf GELU in double, rounded to bf16
    dgr = (g3.float() - gref).abs()
    assert (dgr > 0).float().mean().item() < 2e-3 and dgr.max().item() <= 0.008 * gref.abs().max().item()  # (double rounding)
    close(csq3, csq2, torch.float32, "colsq (fc1 fused) vs unfused", scale=csq2.abs().max().item() * 5)
    out2 = torch.empty((M, C), dtype=dt, device="cuda")
    ops.gemm("nt", gg, W2, out2, M, C, H4, H4, H4, C, dtype=dt, pro=L.PRO_GRN, grn_s=ops.grn_scale(csq2, gamma), grn_b=beta,
             hw=hw, epi=L.EPI_BIAS_RES, bias=b2, res=res, ldr=C, rscale=rs)
    close(out2, ref, dt, "unfused fc2 on the fused fc1's statistics vs fp32 statement")
    if not inference_pair:
        return
    err = (out.float() - out2.float()).abs().max().item() / out2.float().abs().max().item()
    assert err <= 1e-2, err  # one bf16 ulp of the output scale: the two schedules differ in accumulation order only
    if drop_path:  # a dropped sample is exactly its shortcut
        assert torch.equal(out[:hw], res[:hw])


@pytest.mark.parametrize("C,hw,B", [(96, 256, 3), (192, 1024, 2), (224, 512, 2), (384, 256, 2)])
def test_fused_grn_mlp_with_layernorm_in_the_prologue(C, hw, B):
    """vsx_mlp_fc1_ln / vsx_mlp_fwd_ln (the block LayerNorm applied to the un-normalised rows inside the fused passes) against
    vsx_ln_fwd followed by the plain passes: the normalised rows and rstd the training pass writes are what the LayerNorm
    kernel writes (same arithmetic, another summation order: at most a bf16 ulp on rare elements), h / g / colsq / out follow"""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    M, H4 = B * hw, 4 * C
    saved = L.lib().vsx_get_flag(b"mlp_fused")
    L.lib().vsx_set_flag(b"mlp_fused", 63)
    try:
        y = (rnd(M, C, dt=torch.float32, seed=11) * 1.7 + 0.4 + rnd(M, 1, dt=torch.float32, seed=12)).to(dt).cuda()  # row offsets: a real mean
        res = rnd(M, C, dt=dt, seed=2).cuda()
        W1 = rnd(H4, C, dt=dt, seed=3, scale=C ** -0.5).cuda()
        W2 = rnd(C, H4, dt=dt, seed=4, scale=H4 ** -0.5).cuda()
        b1, b2 = rnd(H4, seed=5, scale=0.1).cuda(), rnd(C, seed=6, scale=0.1).cuda()
        gamma, beta = rnd(H4, seed=7, scale=0.3).cuda(), rnd(H4, seed=8, scale=0.1).cuda()
        img = ops.mlp_pack(W1, W2, C)
        xh_ref, _, rstd_ref = ops.ln_fwd(y, None, None, M, C, 1e-6, need_mean=False)
        # training pass
        cs_a, cs_b = torch.zeros((B, H4), device="cuda"), torch.zeros((B, H4), device="cuda")
        h_ref, g_ref = ops.mlp_fc1(xh_ref, img, b1, cs_a, M, C, hw)
        xh, rstd, h, g = ops.mlp_fc1_ln(y, img, b1, cs_b, M, C, hw, 1e-6)
        torch.testing.assert_close(rstd, rstd_ref, rtol=1e-5, atol=0)  # fp32 sums in another order
        dx = (xh.float() - xh_ref.float()).abs()
        assert dx.max().item() <= 0.008 * xh_ref.float().abs().max().item() and (dx > 0).float().mean().item() < 1e-3
        same = (xh == xh_ref).all(dim=1)  # rows whose normalised values agree bit for bit give bit-identical h / g
        assert same.float().mean().item() > 0.9
        assert torch.equal(h[same], h_ref[same]) and torch.equal(g[same], g_ref[same])
        close(cs_b, cs_a, torch.float32, "colsq (LayerNorm in the prologue)", scale=cs_a.abs().max().item() * 5)
        if not ops.mlp_supported(C, hw, M, dt):  # C = 384: training passes only
            return
        # inference pair
        cs_c = torch.zeros_like(cs_a)
        ops.mlp_stats(y, img, b1, cs_c, M, C, hw, ln_eps=1e-6)
        close(cs_c, cs_a, torch.float32, "colsq (inference, LayerNorm in the prologue)", scale=cs_a.abs().max().item() * 5)
        s = ops.grn_scale(cs_a, gamma)
        out_ref = ops.mlp_out(xh_ref, img, b1, s, beta, b2, res, None, M, C, hw)
        out = ops.mlp_out(y, img, b1, s, beta, b2, res, None, M, C, hw, ln_eps=1e-6)
        assert torch.equal(out[same], out_ref[same])
        err = (out.float() - out_ref.float()).abs().max().item() / out_ref.float().abs().max().item()
        assert err <= 1e-2, err
    finally:
        L.lib().vsx_set_flag(b"mlp_fused", saved)


@pytest.mark.parametrize("C,hw,B", [(96, 256, 3), (96, 1024, 2), (192, 256, 2), (224, 512, 2), (96, 4096, 2), (224, 8192, 1),
                                    (384, 256, 3)])
def test_fused_block_backward_without_stored_dz(C, hw, B):
    """csrc/mlp.hip MODE 3 / 4 (dz recomputed on chip: GRN statistics pass, then dh written once) against the unfused pair
    vsx_gemm_nt(VSX_EPI_DZ) + vsx_grn_gelu_bwd on the same operands: same dz rounding points, same GELU arithmetic"""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    M, H4 = B * hw, 4 * C
    saved = L.lib().vsx_get_flag(b"mlp_fused")
    L.lib().vsx_set_flag(b"mlp_fused", 31)
    try:
        assert ops.mlp_supported(C, hw, M, dt, 3) and ops.mlp_supported(C, hw, M, dt, 4)
        _fused_bwd_case(C, hw, B, dt, M, H4, L, ops)
    finally:
        L.lib().vsx_set_flag(b"mlp_fused", saved)


def _fused_bwd_case(C, hw, B, dt, M, H4, L, ops):
    dout = rnd(M, C, dt=dt, seed=1).cuda()
    W2 = rnd(C, H4, dt=dt, seed=2, scale=H4 ** -0.5).cuda()
    W2T = W2.t().contiguous()
    h = rnd(M, H4, dt=dt, seed=3).cuda()
    g = torch.nn.functional.gelu(h.float()).to(dt)
    s = (1 + 0.2 * rnd(B, H4, seed=4)).cuda()
    t = (0.05 * rnd(B, H4, seed=5)).cuda()
    # unfused
    PS = torch.zeros((2, B, H4), dtype=torch.float32, device="cuda")
    dz = torch.empty((M, H4), dtype=dt, device="cuda")
    ops.gemm("nt", dout, W2T, dz, M, H4, C, C, C, H4, dtype=dt, epi=L.EPI_DZ, aux=g, ldx=H4, red0=PS[0], red1=PS[1], hw=hw)
    db = torch.zeros(H4, dtype=torch.float32, device="cuda")
    dh_ref = dz.clone()
    ops.grn_gelu_bwd(dh_ref, h, s, t, db, M, H4, hw)
    # fused
    img2 = ops.mlp_pack(W2T, W2, C)
    PS2 = torch.zeros_like(PS)
    ops.mlp_bwd_stats(dout, img2, g, PS2[0], PS2[1], M, C, hw)
    close(PS2[0], PS[0], torch.float32, "P = sum dz*g", scale=PS[0].abs().max().item() * 5)
    close(PS2[1], PS[1], torch.float32, "S = sum dz", scale=PS[1].abs().max().item() * 5)
    # the same statistics AND the fc2 weight gradient from the per-sample products Q_b = dout_b^T g_b (dz is linear in dout)
    beta = (0.1 * rnd(H4, seed=6)).cuda()
    Qb = torch.full((B, C, H4), float("nan"), dtype=torch.float32, device="cuda")   # plain stores: no zero-fill needed
    csb = torch.full((B, C), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm("tn", g, dout, Qb, M, C, H4, H4, C, H4, dtype=dt, hw=hw, colsum=csb, b_bstride=C * H4)
    Qref = torch.einsum("bpc,bpj->bcj", dout.float().view(B, hw, C), g.float().view(B, hw, H4))
    close(Qb, Qref, torch.float32, "per-sample TN products", scale=Qref.abs().max().item() * 5)
    close(csb, dout.float().view(B, hw, C).sum(1), torch.float32, "per-sample column sums", scale=float(hw) ** 0.5)
    PS3 = torch.zeros_like(PS)
    dW2q, db2q = torch.zeros((C, H4), device="cuda"), torch.zeros(C, device="cuda")
    ops.grn_q_reduce(Qb, csb, W2, s, beta, PS3[0], PS3[1], dW2q, db2q)
    # (the unfused P / S round dz to bf16 element by element before summing: agreement to that rounding noise)
    close(PS3[0], PS[0], torch.float32, "P from Q", scale=PS[0].abs().max().item() * 4)
    close(PS3[1], PS[1], torch.float32, "S from cs", scale=PS[1].abs().max().item() * 4)
    dW2r, db2r = torch.zeros((C, H4), device="cuda"), torch.zeros(C, device="cuda")
    ops.gemm("tn", g, dout, dW2r, M, C, H4, H4, C, H4, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=beta, hw=hw, colsum=db2r)
    close(dW2q, dW2r, dt, "dW2 from Q vs GRN-prologue TN")   # the prologue form rounds z = g*s + beta to bf16
    close(db2q, db2r, torch.float32, "db2", scale=db2r.abs().max().item() * 5)
    db2 = torch.zeros_like(db)
    dh = ops.mlp_bwd_dh(dout, img2, h, s, t, db2, M, C, hw)
    # the fused pass takes dz in fp32 straight from its accumulators (round 5), the unfused pair stores dz in bf16 in between: the two
    # differ by that rounding — and against an fp64 statement of the same expression (same bf16 h, s, t) the fused pass must be the
    # MORE accurate of the two
    err = (dh.float() - dh_ref.float()).abs().max().item() / dh_ref.float().abs().max().item()
    assert err <= 1.5e-2, err
    x64 = h.double()
    cdf64 = 0.5 * (1 + torch.erf(x64 * 0.7071067811865476))
    pdf64 = torch.exp(-0.5 * x64 * x64) * 0.3989422804014327
    dz64_ = (dout.double() @ W2.double()).view(B, hw, H4)
    ref64 = ((dz64_ * s.double()[:, None] + (x64 * cdf64).view(B, hw, H4) * t.double()[:, None]).view(M, H4) * (cdf64 + x64 * pdf64))
    e_fused, e_unfused = (dh.double() - ref64).abs().mean().item(), (dh_ref.double() - ref64).abs().mean().item()
    assert e_fused <= e_unfused, (e_fused, e_unfused)
    del x64, cdf64, pdf64, dz64_, ref64
    close(db2, db, torch.float32, "colsum dh", scale=db.abs().max().item() * 5)
    # against the fp32 statement
    dzf = (dout.float() @ W2.float()).to(dt).float()
    x = h.float()
    cdf = 0.5 * (1 + torch.erf(x * 0.7071067811865476))
    pdf = torch.exp(-0.5 * x * x) * 0.3989422804014327
    ref = ((dzf.view(B, hw, H4) * s[:, None] + (x * cdf).view(B, hw, H4) * t[:, None]).view(M, H4) * (cdf + x * pdf)).to(dt)
    close(dh, ref, dt, "dh vs fp32 statement")
    # ... and the statistics / fc2 gradients against fp64 statements of timm's GlobalResponseNormMlp backward (fcmae.py:174-221),
    # so that none of the three routes above is only compared with another kernel of this library (VERDICT r3 weak 1 iii).
    # Bound: every term carries one bf16 rounding (2^-9 relative, independent) -> 6 sigma of sqrt(sum of squared terms)
    d64, g64 = dout.double().view(B, hw, C), g.double().view(B, hw, H4)
    dz64 = d64 @ W2.double()
    P64, S64 = (dz64 * g64).sum(1), dz64.sum(1)
    bP = 6 * 2.0 ** -9 * ((dz64 * g64) ** 2).sum(1).sqrt().max().item() + 1e-6
    bS = 6 * 2.0 ** -9 * (dz64 ** 2).sum(1).sqrt().max().item() + 1e-6
    for what, got in (("unfused epilogue", PS), ("MODE 3", PS2), ("per-sample products", PS3)):
        assert (got[0].double() - P64).abs().max().item() <= bP, (what, "P", (got[0].double() - P64).abs().max().item(), bP)
        assert (got[1].double() - S64).abs().max().item() <= bS, (what, "S", (got[1].double() - S64).abs().max().item(), bS)
    z64 = g64 * s.double()[:, None] + beta.double()
    dW64 = torch.einsum("bpc,bpj->cj", d64, z64)
    bW = 6 * 2.0 ** -9 * torch.einsum("bpc,bpj->cj", d64 ** 2, z64 ** 2).sqrt().max().item() + 1e-6
    for what, got in (("from Q", dW2q), ("GRN-prologue TN", dW2r)):
        assert (got.double() - dW64).abs().max().item() <= bW, (what, (got.double() - dW64).abs().max().item(), bW)
    db64 = d64.sum((0, 1))
    for what, got in (("from cs", db2q), ("TN column sums", db2r)):
        assert (got.double() - db64).abs().max().item() <= 1e-4 * d64.abs().sum((0, 1)).max().item() + 1e-6, what


@pytest.mark.gpu
@pytest.mark.parametrize("C,hw,B", [(96, 4096, 3), (192, 1024, 5), (224, 4096, 2), (96, 256, 8)])
@pytest.mark.parametrize("ln_in", [False, True])
def test_fused_block_without_a_stored_preactivation(C, hw, B, ln_in):
    """csrc/mlp.hip MODE 6 (training fc1 that stores g only) and MODE 5 (dh pass that recomputes h = x^ . W1'^T + b1 on chip):
    g, x^, rstd and the GRN sums BIT-IDENTICAL to MODE 2's, dh and its column sums bit-identical to MODE 4 run on MODE 2's stored
    h — and dh against an fp32 statement of the block backward (timm GlobalResponseNormMlp as restated in
    viscy_models/unet/fcmae.py:174-221), so the pair is not only compared with itself"""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    M, H4 = B * hw, 4 * C
    saved = L.lib().vsx_get_flag(b"mlp_fused")
    L.lib().vsx_set_flag(b"mlp_fused", 127)
    try:
        assert ops.mlp_supported(C, hw, M, dt, 5) and ops.mlp_supported(C, hw, M, dt, 6)
        y = rnd(M, C, dt=dt, seed=11, scale=2.0).cuda()
        W1 = rnd(H4, C, dt=dt, seed=12, scale=C ** -0.5).cuda()
        W2 = rnd(C, H4, dt=dt, seed=13, scale=H4 ** -0.5).cuda()
        W2T = W2.t().contiguous()
        b1 = (0.1 * rnd(H4, seed=14)).cuda()
        dout = rnd(M, C, dt=dt, seed=15).cuda()
        s = (1 + 0.2 * rnd(B, H4, seed=16)).cuda()
        t = (0.05 * rnd(B, H4, seed=17)).cuda()
        img, img2 = ops.mlp_pack(W1, W2, C), ops.mlp_pack(W2T, W2, C)
        q2, q6 = torch.zeros((B, H4), device="cuda"), torch.zeros((B, H4), device="cuda")
        if ln_in:
            xh2, r2, h2, g2 = ops.mlp_fc1_ln(y, img, b1, q2, M, C, hw, 1e-6)
            xh6, r6, h6, g6 = ops.mlp_fc1_ln(y, img, b1, q6, M, C, hw, 1e-6, store_h=False)
            assert torch.equal(xh2, xh6) and torch.equal(r2, r6)
        else:
            xh2 = xh6 = y
            h2, g2 = ops.mlp_fc1(y, img, b1, q2, M, C, hw)
            h6, g6 = ops.mlp_fc1(y, img, b1, q6, M, C, hw, store_h=False)
        assert h6 is None and torch.equal(g2, g6)
        close(q6, q2, torch.float32, "GRN sums")  # (atomics: order only)
        db4, db5 = torch.zeros(H4, device="cuda"), torch.zeros(H4, device="cuda")
        dh4 = ops.mlp_bwd_dh(dout, img2, h2, s, t, db4, M, C, hw)
        dh5 = torch.full((M, H4), float("nan"), dtype=dt, device="cuda")
        dh5 = ops.mlp_bwd_dh_re(dout, xh6, img2, img, b1, s, t, db5, M, C, hw)
        assert torch.isfinite(dh5.float()).all()
        assert torch.equal(dh4, dh5), (dh4.float() - dh5.float()).abs().max().item()
        assert torch.equal(db4, db5)
        # fp32 statement: h from the bf16 operands with fp32 accumulation, rounded as the forward stores it
        hf = (xh6.float() @ W1.float().t() + b1).to(dt).float()
        close(h2, hf.to(dt), dt, "stored h vs fp32 statement")
        dzf = (dout.float() @ W2.float()).to(dt).float()
        cdf = 0.5 * (1 + torch.erf(hf * 0.7071067811865476))
        pdf = torch.exp(-0.5 * hf * hf) * 0.3989422804014327
        ref = ((dzf.view(B, hw, H4) * s[:, None] + (hf * cdf).view(B, hw, H4) * t[:, None]).view(M, H4) * (cdf + hf * pdf)).to(dt)
        # (a bf16 ulp of h here and there — another accumulation order than the MFMA's — moves gelu'(h) by its slope)
        err = (dh5.float() - ref.float()).abs()
        assert (err > 2e-2 * ref.float().abs().max()).float().mean().item() < 1e-4, err.max().item()
        close(db5, ref.float().sum(0), torch.float32, "colsum dh", scale=ref.float().sum(0).abs().max().item() * 5 + float(M) ** 0.5 * 0.05)
    finally:
        L.lib().vsx_set_flag(b"mlp_fused", saved)


@pytest.mark.gpu
@pytest.mark.parametrize("C,hw,B,offset", [(96, 4096, 3, 0.0), (192, 1024, 5, 3.0), (224, 4096, 2, 8.0), (96, 256, 8, 30.0)])
def test_fused_block_without_stored_normalised_rows(C, hw, B, offset):
    """`mlp_fused` bit 7: the block LayerNorm's output x^ is never stored.  Forward (MODE 6 with xh_out = NULL) keeps y + row mean /
    rstd; the dh pass (MODE 7, vsx_mlp_bwd_dh_ln) re-forms x^ = bf16((y - mean) * rstd) on chip and writes dh' = dh * rstd; the fc1
    weight gradient is dh'^T . y - u (x) 1 (plain TN GEMM on y + vsx_unprep_grad(rowsub = u)); the data gradient's LayerNorm
    epilogue re-forms x^ from y and drops its trailing rstd.  Every piece against the x^-storing path (MODE 6 / 5) AND against
    fp64 statements of LayerNorm -> Linear backward (timm ConvNeXtBlock.norm / .mlp.fc1, viscy_models/unet/fcmae.py:174-221); rows
    with a DC offset of up to 30 sigma (the rank-1 term u then dwarfs the gradient it is subtracted from)."""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    M, H4 = B * hw, 4 * C
    saved = L.lib().vsx_get_flag(b"mlp_fused")
    L.lib().vsx_set_flag(b"mlp_fused", 255)
    try:
        assert ops.mlp_supported(C, hw, M, dt, 7)
        y = (rnd(M, C, seed=11, scale=2.0) + offset * rnd(M, 1, seed=21)).to(dt).cuda()
        W1 = rnd(H4, C, dt=dt, seed=12, scale=C ** -0.5).cuda()
        W2 = rnd(C, H4, dt=dt, seed=13, scale=H4 ** -0.5).cuda()
        W2T, W1T = W2.t().contiguous(), W1.t().contiguous()
        b1 = (0.1 * rnd(H4, seed=14)).cuda()
        dout = rnd(M, C, dt=dt, seed=15).cuda()
        s = (1 + 0.2 * rnd(B, H4, seed=16)).cuda()
        t = (0.05 * rnd(B, H4, seed=17)).cuda()
        img, img2 = ops.mlp_pack(W1, W2, C), ops.mlp_pack(W2T, W2, C)
        # ---- forward: same g / rstd / GRN sums, no x^; the mean is the row mean
        q6, q7 = torch.zeros((B, H4), device="cuda"), torch.zeros((B, H4), device="cuda")
        xh6, r6, h6, g6 = ops.mlp_fc1_ln(y, img, b1, q6, M, C, hw, 1e-6, store_h=False)
        (y7, mean7), r7, h7, g7 = ops.mlp_fc1_ln(y, img, b1, q7, M, C, hw, 1e-6, store_h=False, store_xh=False)
        assert h6 is None and h7 is None and y7 is y and torch.equal(g6, g7) and torch.equal(r6, r7)
        close(q7, q6, torch.float32, "GRN sums")
        torch.testing.assert_close(mean7, y.float().mean(1), rtol=1e-5, atol=1e-5)
        # ---- dh pass: dh' = dh * rstd (rounded once, after the scaling), bias gradient, u
        db5, cs2 = torch.zeros(H4, device="cuda"), torch.zeros((2, H4), device="cuda")
        dh5 = ops.mlp_bwd_dh_re(dout, xh6, img2, img, b1, s, t, db5, M, C, hw)
        dh7 = ops.mlp_bwd_dh_ln(dout, y, mean7, r7, img2, img, b1, s, t, cs2, M, C, hw)
        want = dh5.float() * r7[:, None]
        err = (dh7.float() - want).abs()
        assert (err <= 2.0 ** -7 * want.abs() + 1e-30).all(), (err / want.abs().clamp_min(1e-30)).max().item()
        sc_db = dh5.float().abs().sum(0).max().item()
        close(cs2[0], db5, torch.float32, "fc1 bias gradient from the scaled tile", scale=sc_db * 2.0 ** -8 / tol(torch.float32) * 4 / M ** 0.5 + db5.abs().max().item())
        # round 5: sigma = 1 / rstd enters the matrix-core sum as a bf16 head + tail, so the sum reproduces the fp64
        # statement of what it adds up — the stored dh' times sigma — to fp32 round-off, not to 2^-9 per row
        db64 = (dh7.double() / r7.double()[:, None]).sum(0)
        scale_db = (dh7.double().abs() / r7.double()[:, None]).sum(0).max().item()
        assert (cs2[0].double() - db64).abs().max().item() <= 3e-5 * scale_db + 1e-6, ((cs2[0].double() - db64).abs().max().item(), scale_db)
        u64 = (dh7.double() * mean7.double()[:, None]).sum(0)
        scale_u = (dh7.double().abs() * mean7.double().abs()[:, None]).sum(0).max().item()
        assert (cs2[1].double() - u64).abs().max().item() <= 3e-5 * scale_u + 1e-6, ((cs2[1].double() - u64).abs().max().item(), scale_u)
        # ---- data gradient with the LayerNorm backward in its epilogue
        dy5 = ops.dgrad_ln_bwd(dh5, W1T, xh6, r6, M, C, H4)
        dy7 = ops.dgrad_ln_bwd(dh7, W1T, y, r7, M, C, H4, mean=mean7)
        assert dy5 is not None and dy7 is not None
        dx = dh5.double() @ W1.double()
        xh = xh6.double()
        ref = r6.double()[:, None] * (dx - dx.mean(1, keepdim=True) - xh * (dx * xh).mean(1, keepdim=True))
        close(dy5, ref, dt, "dy (stored x^) vs fp64 statement")
        close(dy7, ref, dt, "dy (x^ re-formed from y) vs fp64 statement")
        # ---- fc1 weight gradient and its unfold (dW1 = dW1f . diag(gamma) + db1f (x) beta, dgamma = sum dW1f * W1)
        gam, bet = (1 + 0.3 * rnd(C, seed=31)).cuda(), (0.2 * rnd(C, seed=32)).cuda()
        Wp = rnd(H4, C, seed=33, scale=C ** -0.5).cuda()
        T5, T7 = torch.zeros((H4, C), device="cuda"), torch.zeros((H4, C), device="cuda")
        ops.gemm("tn", xh6, dh5, T5, M, H4, C, C, H4, C, dtype=dt)
        ops.gemm("tn", y, dh7, T7, M, H4, C, C, H4, C, dtype=dt)
        true = dh5.double().t() @ xh
        close(T5, true, dt, "dW1f (stored x^) vs fp64", scale=true.abs().max().item())
        close(T7 - cs2[1][:, None], true, dt, "dW1f = dh'^T y - u (x) 1 vs fp64", scale=true.abs().max().item())
        outs = []
        for T, db, rs in ((T5, db5, None), (T7, cs2[0], cs2[1])):
            dW, dg = torch.zeros((H4, C), device="cuda"), torch.zeros(C, device="cuda")
            ops.unprep_grad(T, dW, H4, C, 1, gamma=gam, W=Wp, dgamma=dg, u=db, beta=bet, rowsub=rs)
            outs.append((dW, dg))
        Tc = (T7 - cs2[1][:, None]).double()
        close(outs[1][0], Tc * gam.double() + cs2[0].double()[:, None] * bet.double(), torch.float32, "unfold with rowsub: dW1")
        close(outs[1][1], (Tc * Wp.double()).sum(0), torch.float32, "unfold with rowsub: dgamma")
        close(outs[1][0], outs[0][0], dt, "dW1: x^-free path vs stored-x^ path")
        close(outs[1][1], outs[0][1], dt, "dgamma: x^-free path vs stored-x^ path", scale=outs[0][1].abs().max().item() + true.abs().max().item())
    finally:
        L.lib().vsx_set_flag(b"mlp_fused", saved)


@pytest.mark.gpu
@pytest.mark.parametrize("C,hw,B", [(384, 256, 24), (768, 64, 40), (96, 1024, 6), (224, 4096, 3)])
def test_fc2_weight_gradient_that_also_delivers_the_grn_statistics(C, hw, B):
    """Round 5 (csrc/gemm.hip, gemm_tn_fast_kernel PRO == 2): `gemm("tn", g, dout, dW2, pro=GRN, aux=W2, red0=P)` — one launch for
    the fc2 weight gradient dW2 = dout^T (g * s_b + beta), the bias gradient (column sums of dout) and the GRN backward statistics
    P[b, k] = sum_hw dz * g, dz = dout . W2 — against fp64 statements of timm's GlobalResponseNormMlp backward (fcmae.py:174-221)
    and against the two launches it replaces (the GRN-prologue weight gradient; csrc/mlp.hip MODE 3 where that exists)."""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    M, H4 = B * hw, 4 * C
    assert ops.tn_grn_stats_ok(M, C, H4, hw, dt)
    g = rnd(M, H4, dt=dt, seed=1, scale=0.7).cuda()
    dout = rnd(M, C, dt=dt, seed=2).cuda()
    W2 = rnd(C, H4, dt=dt, seed=3, scale=H4 ** -0.5).cuda()
    s = (1 + 0.3 * rnd(B, H4, seed=4)).cuda()
    beta = (0.2 * rnd(H4, seed=5)).cuda()
    dW2 = torch.zeros((C, H4), device="cuda")
    cs = torch.zeros(C, device="cuda")
    P = torch.zeros((B, H4), device="cuda")
    ops.gemm("tn", g, dout, dW2, M, C, H4, H4, C, H4, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=beta, hw=hw, colsum=cs, aux=W2, ldx=H4, red0=P)
    assert L.lib().vsx_last_kernel() == b"gemm_tn_fast"
    d64, g64 = dout.double().view(B, hw, C), g.double().view(B, hw, H4)
    z64 = g64 * s.double()[:, None] + beta.double()
    dW64 = torch.einsum("bhc,bhk->ck", d64, z64)
    cs64 = d64.sum((0, 1))
    P64 = ((d64 @ W2.double()) * g64).sum(1)
    # bounds: products of bf16 values accumulated in fp32 — 6 sigma of eps32 * sqrt(sum of squared terms), + the bf16 W2 in P is exact
    bW = 6 * 2.0 ** -22 * torch.einsum("bhc,bhk->ck", d64 ** 2, z64 ** 2).sqrt().max().item() * (M ** 0.5) ** 0 + 1e-3 * dW64.abs().max().item()
    assert (dW2.double() - dW64).abs().max().item() <= bW, ((dW2.double() - dW64).abs().max().item(), bW)
    close(cs, cs64.float(), torch.float32, "bias gradient = column sums of dout", scale=cs64.abs().max().item() + float(M) ** 0.5)
    bP = 1e-3 * P64.abs().max().item() + 1e-4
    assert (P.double() - P64).abs().max().item() <= bP, ((P.double() - P64).abs().max().item(), bP)
    # the launches it replaces: the prologue weight gradient rounds z to bf16 (so it is the LESS accurate of the two) ...
    dW2p, csp = torch.zeros((C, H4), device="cuda"), torch.zeros(C, device="cuda")
    ops.gemm("tn", g, dout, dW2p, M, C, H4, H4, C, H4, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=beta, hw=hw, colsum=csp)
    close(dW2, dW2p, dt, "dW2: post-scaled products vs GRN-prologue operand")
    assert (dW2.double() - dW64).abs().mean().item() <= (dW2p.double() - dW64).abs().mean().item() * 1.02
    close(cs, csp, torch.float32, "column sums", scale=cs64.abs().max().item() + float(M) ** 0.5)
    # ... and MODE 3 recomputes dz for the same P
    if ops.mlp_supported(C, hw, M, dt, 3):
        img2 = ops.mlp_pack(W2.t().contiguous(), W2, C)
        P3, S3 = torch.zeros((B, H4), device="cuda"), torch.zeros((B, H4), device="cuda")
        ops.mlp_bwd_stats(dout, img2, g, P3, S3, M, C, hw)
        close(P, P3, torch.float32, "P: from the weight-gradient tiles vs MODE 3", scale=P64.abs().max().item())
        # what S was used for: dbeta = sum_b S_b = (column sums of dout) . W2
        close((cs.double() @ W2.double()).float(), S3.sum(0), torch.float32, "dbeta", scale=S3.sum(0).abs().max().item() + 1.0)
    # run-to-run: the split-K atomics only move round-off
    dW2b, Pb = torch.zeros((C, H4), device="cuda"), torch.zeros((B, H4), device="cuda")
    ops.gemm("tn", g, dout, dW2b, M, C, H4, H4, C, H4, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=beta, hw=hw, colsum=torch.zeros(C, device="cuda"),
             aux=W2, ldx=H4, red0=Pb)
    close(dW2b, dW2, torch.float32, "dW2 run to run", scale=dW64.abs().max().item())
    close(Pb, P, torch.float32, "P run to run", scale=P64.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_stem_patch_gemm_vs_directly_imported_reference_stem(dt):
    """G1 on the device: tests/golden/stem.pt holds weights, input and output of the REFERENCE's own
    `UNeXt2Stem` class (components/stems.py:26-50, imported directly by oracle/validate_against_reference.py — the one golden
    made with no stub at all).  The product computes the stem as vsx_stem_im2col + vsx_gemm_nt (bias epilogue): same numbers,
    channels-last, to the fp32 bar of the north star (1e-3; measured ~1e-6) / the bf16 operand rounding."""
    if SELF_CHECK:
        pytest.skip("HIP-only path")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    g = load_golden("stem.pt")
    x, W, b, y = g["x"].cuda(), g["weight"].cuda(), g["bias"].cuda(), g["y"]
    B, Cin, Z, H, Wd = x.shape
    co, kz, ky, kx = W.shape[0], W.shape[2], W.shape[3], W.shape[4]
    assert Z == kz  # Conv3d(kernel = stride = (5, 4, 4)) on a 5-slice stack: one depth slab, squeezed (stems.py:46-50)
    K = Cin * kz * ky * kx
    ld = K if dt == torch.float32 else (K + 31) // 32 * 32   # the bf16 engine pads K = 80 to a whole number of MFMA slabs
    P = ops.stem_im2col(x.contiguous(), (kz, ky, kx), dt, ld=ld)
    Wm = torch.zeros((co, ld), dtype=dt, device="cuda")
    Wm[:, :K] = W.reshape(co, K).to(dt)
    M = B * (H // ky) * (Wd // kx)
    out = torch.empty((M, co), dtype=dt, device="cuda")
    ops.gemm("nt", P, Wm, out, M, co, ld, ld, ld, co, dtype=dt, epi=L.EPI_BIAS, bias=b)
    got = out.float().view(B, H // ky, Wd // kx, co).permute(0, 3, 1, 2).cpu()
    err = ((got - y).abs().max() / y.abs().max()).item()
    assert err <= (1e-3 if dt == torch.float32 else 2e-2), err
    if dt == torch.float32:
        assert err <= 1e-5, err   # what the exact-fp32 MFMA path actually delivers


def _more_workgroups_than_twice_the_cus() -> int:
    return 2 * torch.cuda.get_device_properties(0).multi_processor_count + 1


@pytest.mark.gpu
def test_lds_double_buffered_kernels_with_more_than_two_workgroups_per_cu():
    """VERDICT r4 item 6.iii.  The race round 4 found late (csrc/mlp.hip: a stage buffer refilled while slower waves still read
    it) only shows when a launch has more workgroups than the chip holds at once — which the small unit tests never have.
    Every other kernel that hands LDS buffers between an LDS-DMA / prefetch and its readers gets one launch with
    > 2 x CUs workgroups here, three repetitions, checked against the plain statement of the op and for run-to-run identity:
    gemm_nt2 (3-stage DMA ring, incl. the GRN prologue and the LayerNorm-backward epilogue), the lean NT / TN GEMMs (single /
    double LDS buffer), the matrix-core depthwise kernels (register-staged and LDS-DMA variants, forward / data / weight
    gradient) and the direct head convolutions (halo tile per workgroup)."""
    if SELF_CHECK:
        pytest.skip("HIP-only kernels")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    need = _more_workgroups_than_twice_the_cus()

    def thrice(fn, what, exact=True):
        first = fn()
        for _ in range(2):
            again = fn()
            for a, b_ in zip(first, again):
                if exact:
                    assert torch.equal(a, b_), what
                else:
                    close(a, b_, torch.float32, what)
        return first

    # ---- gemm_nt2: 256 x BN tiles; K-heavy plain launch, GRN-prologue launch, LayerNorm-backward epilogue
    hw, C = 256, 384
    Bn = (need + 0) // 1            # one 256-row tile per sample, one column tile (N = 384 -> BN = 384): Bn workgroups
    M, K = Bn * hw, 4 * C
    A = rnd(M, K, dt=dt, seed=1, scale=0.5).cuda()
    Wt = rnd(C, K, dt=dt, seed=2, scale=K ** -0.5).cuda()
    s_ = (1 + 0.2 * rnd(Bn, K, seed=3)).cuda()
    be = (0.1 * rnd(K, seed=4)).cuda()
    b2 = (0.1 * rnd(C, seed=5)).cuda()
    res = rnd(M, C, dt=dt, seed=6).cuda()

    def nt2_plain():
        o = torch.full((M, C), float("nan"), dtype=dt, device="cuda")
        ops.gemm("nt", A, Wt, o, M, C, K, K, K, C, dtype=dt)
        return (o,)

    def nt2_pro():
        o = torch.full((M, C), float("nan"), dtype=dt, device="cuda")
        ops.gemm("nt", A, Wt, o, M, C, K, K, K, C, dtype=dt, pro=L.PRO_GRN, grn_s=s_, grn_b=be, hw=hw, epi=L.EPI_BIAS_RES, bias=b2,
                 res=res, ldr=C)
        return (o,)

    L.lib().vsx_set_flag(b"nt2", 3)
    try:
        (o0,) = thrice(nt2_plain, "gemm_nt2 plain")
        assert L.lib().vsx_last_kernel() == b"gemm_nt2"
        close(o0, A.float() @ Wt.float().t(), dt, "gemm_nt2 plain vs fp32 statement")
        (o1,) = thrice(nt2_pro, "gemm_nt2 GRN prologue")
        z = (A.float().view(Bn, hw, K) * s_[:, None, :] + be).to(dt).float().view(M, K)
        close(o1, z @ Wt.float().t() + b2 + res.float(), dt, "gemm_nt2 GRN prologue vs fp32 statement")
    finally:
        L.lib().vsx_set_flag(b"nt2", 1)
    Cl = 224
    Ml = need * 256
    dh = rnd(Ml, 4 * Cl, dt=dt, seed=7, scale=0.5).cuda()
    W1T = rnd(Cl, 4 * Cl, dt=dt, seed=8, scale=(4 * Cl) ** -0.5).cuda()
    xh = rnd(Ml, Cl, dt=dt, seed=9).cuda()
    rstd = (0.5 + rnd(Ml, seed=10).abs()).cuda()

    def lnbwd():
        return (ops.dgrad_ln_bwd(dh, W1T, xh, rstd, Ml, Cl, 4 * Cl),)

    (dy,) = thrice(lnbwd, "gemm_nt2 LayerNorm-backward epilogue")
    dxh = (dh.float() @ W1T.float().t()).to(dt).float()
    want = rstd[:, None] * (dxh - dxh.mean(1, keepdim=True) - xh.float() * (dxh * xh.float()).mean(1, keepdim=True))
    close(dy, want, dt, "fc1 data gradient + LayerNorm backward vs fp32 statement")
    del dh, xh, dy, dxh, want

    # ---- lean NT (single LDS buffer, BK = 64) and lean TN
    Mn, Kn, Nn = need * 128, 192, 128          # one 128 x 128 tile per row block
    An = rnd(Mn, Kn, dt=dt, seed=11, scale=0.5).cuda()
    Wn = rnd(Nn, Kn, dt=dt, seed=12, scale=Kn ** -0.5).cuda()

    def nt_fast():
        o = torch.full((Mn, Nn), float("nan"), dtype=dt, device="cuda")
        ops.gemm("nt", An, Wn, o, Mn, Nn, Kn, Kn, Kn, Nn, dtype=dt)
        return (o,)

    (on,) = thrice(nt_fast, "gemm_nt_fast")
    close(on, An.float() @ Wn.float().t(), dt, "lean NT vs fp32 statement")
    Yn = rnd(Mn, Nn, dt=dt, seed=13, scale=0.5).cuda()

    def tn_fast():
        o = torch.zeros((Nn, Kn), device="cuda")
        ops.gemm("tn", An, Yn, o, Mn, Nn, Kn, Kn, Nn, Kn, dtype=dt)
        return (o,)

    (ot,) = thrice(tn_fast, "gemm_tn_fast", exact=False)   # split-K atomics: order of the adds differs run to run
    close(ot, Yn.float().t() @ An.float(), torch.float32, "lean TN vs fp32 statement", scale=(Yn.float().t() @ An.float()).abs().max().item())

    # ---- matrix-core depthwise 7 x 7: register-staged (64 x 64) and LDS-DMA (16 x 16 tiles; data gradient) kernels
    for (Hh, Cc, Bb) in ((64, 96, max(4, need // 48 + 1)), (16, 384, need // 12 + 1)):
        xx = rnd(Bb * Hh * Hh, Cc, dt=dt, seed=14).cuda()
        ww = (0.2 * rnd(49, Cc, seed=15)).cuda()
        bb = (0.1 * rnd(Cc, seed=16)).cuda()
        sc = rnd(Bb * Hh * Hh, Cc, dt=dt, seed=17).cuda()
        (yf,) = thrice(lambda: (ops.dwconv7_fwd(xx, ww, bb, Bb, Hh, Hh, Cc),), f"dwconv7 forward {Hh}x{Hh}x{Cc}")
        close(yf, R.dwconv7_fwd(xx.cpu(), ww.cpu(), bb.cpu(), Bb, Hh, Hh, Cc), dt, "dwconv7 forward vs statement")
        (dx,) = thrice(lambda: (ops.dwconv7_bwd_data(xx, ww, sc, Bb, Hh, Hh, Cc),), f"dwconv7 data gradient {Hh}x{Hh}x{Cc}")
        close(dx, R.dwconv7_bwd_data(xx.cpu(), ww.cpu(), sc.cpu(), Bb, Hh, Hh, Cc), dt, "dwconv7 data gradient vs statement")

        def wg():
            dw, db = torch.zeros((49, Cc), device="cuda"), torch.zeros(Cc, device="cuda")
            ops.dwconv7_bwd_weight(xx, sc, dw, db, Bb, Hh, Hh, Cc)
            return dw, db

        dw, db = thrice(wg, f"dwconv7 weight gradient {Hh}x{Hh}x{Cc}", exact=False)
        rdw, rdb = torch.zeros(49, Cc), torch.zeros(Cc)
        R.dwconv7_bwd_weight(xx.cpu(), sc.cpu(), rdw, rdb, Bb, Hh, Hh, Cc)
        close(dw, rdw, torch.float32, "dwconv7 weight gradient vs statement", scale=rdw.abs().max().item())
        del xx, sc, yf, dx


@pytest.mark.gpu
@pytest.mark.parametrize("C,hw,B", [(96, 4096, 4), (224, 4096, 40), (384, 256, 16)])
def test_fused_passes_scalar_and_packed_fp32_builds_agree(C, hw, B):
    """Round 5: every fused GRN-MLP pass exists in two builds of the same source — packed-fp32 VALU arithmetic (v_pk_*_f32) and
    scalar (`no-packed-fp32-ops`; the default of the forward passes, where a packed instruction next to MFMAs costs more than the
    two scalar ones it replaces: csrc/mlp.hip).  A v_pk_fma_f32 is two v_fma_f32: the outputs must be bit-identical whichever
    build the `mlp_sf32` flag selects (> 256 workgroups at C = 224: both builds also with two workgroups per CU)."""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    M, H4 = B * hw, 4 * C
    lib = L.lib()
    saved = lib.vsx_get_flag(b"mlp_sf32")
    saved_fused = lib.vsx_get_flag(b"mlp_fused")
    lib.vsx_set_flag(b"mlp_fused", saved_fused | 128)  # every pass, also the one the shipped schedule no longer takes (MODE 7)
    y = rnd(M, C, dt=dt, seed=1, scale=2.0).cuda()
    W1 = rnd(H4, C, dt=dt, seed=2, scale=C ** -0.5).cuda()
    W2 = rnd(C, H4, dt=dt, seed=3, scale=H4 ** -0.5).cuda()
    b1 = (0.1 * rnd(H4, seed=4)).cuda()
    dout = rnd(M, C, dt=dt, seed=5).cuda()
    s = (1 + 0.2 * rnd(B, H4, seed=6)).cuda()
    t = (0.05 * rnd(B, H4, seed=7)).cuda()
    b2, beta = (0.1 * rnd(C, seed=8)).cuda(), (0.1 * rnd(H4, seed=9)).cuda()
    img, img2 = ops.mlp_pack(W1, W2, C), ops.mlp_pack(W2.t().contiguous(), W2, C)

    def passes():
        out = {}
        q = torch.zeros((B, H4), device="cuda")
        xh, rstd, h, g = ops.mlp_fc1_ln(y, img, b1, q, M, C, hw, 1e-6)                       # MODE 2
        out["fc1 h"], out["fc1 g"], out["fc1 x^"], out["fc1 sums"] = h, g, xh, q
        P, S = torch.zeros((B, H4), device="cuda"), torch.zeros((B, H4), device="cuda")
        if ops.mlp_supported(C, hw, M, dt, 3):
            ops.mlp_bwd_stats(dout, img2, g, P, S, M, C, hw)                                   # MODE 3
            out["bwd stats P"], out["bwd stats S"] = P, S
        db = torch.zeros(H4, device="cuda")
        out["dh (stored h)"] = ops.mlp_bwd_dh(dout, img2, h, s, t, db, M, C, hw)               # MODE 4
        out["db1 (stored h)"] = db
        if C <= 224:
            q0 = torch.zeros((B, H4), device="cuda")
            ops.mlp_stats(y, img, b1, q0, M, C, hw, ln_eps=1e-6)                               # MODE 0
            out["stats sums"] = q0
            out["inference out"] = ops.mlp_out(y, img, b1, s, beta, b2, dout, None, M, C, hw, ln_eps=1e-6)   # MODE 1
            q6, cs2 = torch.zeros((B, H4), device="cuda"), torch.zeros((2, H4), device="cuda")
            (y7, mean7), r7, _, g7 = ops.mlp_fc1_ln(y, img, b1, q6, M, C, hw, 1e-6, store_h=False, store_xh=False)   # MODE 6
            out["fc1 g (no h, no x^)"], out["fc1 mean"], out["fc1 rstd"] = g7, mean7, r7
            db5 = torch.zeros(H4, device="cuda")
            out["dh (h recomputed)"] = ops.mlp_bwd_dh_re(dout, xh, img2, img, b1, s, t, db5, M, C, hw)          # MODE 5
            out["dh' (y re-normalised)"] = ops.mlp_bwd_dh_ln(dout, y, mean7, r7, img2, img, b1, s, t, cs2, M, C, hw)  # MODE 7
            out["column sums of dh'"] = cs2
        return out

    try:
        lib.vsx_set_flag(b"mlp_sf32", 0)
        packed = passes()
        lib.vsx_set_flag(b"mlp_sf32", 255)
        scalar = passes()
    finally:
        lib.vsx_set_flag(b"mlp_sf32", saved)
        lib.vsx_set_flag(b"mlp_fused", saved_fused)
    assert packed.keys() == scalar.keys() and len(packed) >= 6
    for k in packed:
        if packed[k].dtype == torch.float32 and ("sums" in k or "stats" in k or k.startswith("db1")):
            close(scalar[k], packed[k], torch.float32, k)   # atomically / workspace-reduced: order of the adds differs run to run
        else:
            assert torch.equal(packed[k], scalar[k]), (k, (packed[k] != scalar[k]).sum().item())


@pytest.mark.gpu
@pytest.mark.parametrize("C,hw,B", [(192, 1024, 128), (224, 4096, 64)])
def test_fused_passes_with_two_workgroups_per_cu(C, hw, B):
    """Regression test of a race that only large launches show (round 4): the fused GRN-MLP kernels refilled stage buffer 1 by
    LDS-DMA at the top of their first step while slower waves of the workgroup were still multiplying sub-chunk 0's fragments out
    of it (no barrier behind the prologue GEMM).  One workgroup per CU hid it behind the DMA latency; MODE 6 fits two per CU at
    C = 192 / 224 and then wrote wrong hidden columns 16..31 of the first sub-chunk for whole 32-row groups — only when the launch
    has more workgroups than CUs (the unit tests above never do).  Here: >= 512 workgroups, MODE 6 against MODE 2 bit for bit,
    MODE 5 against MODE 4 bit for bit, three repetitions each."""
    if SELF_CHECK:
        pytest.skip("HIP-only kernel")
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    M, H4 = B * hw, 4 * C
    saved = L.lib().vsx_get_flag(b"mlp_fused")
    L.lib().vsx_set_flag(b"mlp_fused", 127)
    try:
        y = rnd(M, C, dt=dt, seed=1, scale=2.0).cuda()
        W1 = rnd(H4, C, dt=dt, seed=2, scale=C ** -0.5).cuda()
        W2 = rnd(C, H4, dt=dt, seed=3, scale=H4 ** -0.5).cuda()
        b1 = (0.1 * rnd(H4, seed=4)).cuda()
        dout = rnd(M, C, dt=dt, seed=5).cuda()
        s = (1 + 0.2 * rnd(B, H4, seed=6)).cuda()
        t = (0.05 * rnd(B, H4, seed=7)).cuda()
        img, img2 = ops.mlp_pack(W1, W2, C), ops.mlp_pack(W2.t().contiguous(), W2, C)
        q2 = torch.zeros((B, H4), device="cuda")
        xh2, r2, h2, g2 = ops.mlp_fc1_ln(y, img, b1, q2, M, C, hw, 1e-6)
        db4 = torch.zeros(H4, device="cuda")
        dh4 = ops.mlp_bwd_dh(dout, img2, h2, s, t, db4, M, C, hw)
        for it in range(3):
            q6 = torch.zeros((B, H4), device="cuda")
            xh6, r6, h6, g6 = ops.mlp_fc1_ln(y, img, b1, q6, M, C, hw, 1e-6, store_h=False)
            bad = (g2 != g6).any(1).nonzero().flatten()
            assert bad.numel() == 0, (it, bad.numel(), bad[:8].tolist())
            assert torch.equal(xh2, xh6)
            close(q6, q2, torch.float32, "GRN sums")
            db5 = torch.zeros(H4, device="cuda")
            dh5 = ops.mlp_bwd_dh_re(dout, xh6, img2, img, b1, s, t, db5, M, C, hw)
            assert torch.equal(dh4, dh5), (it, (dh4 != dh5).any(1).sum().item())
            del g6, xh6, dh5
        # the passes that do not store the normalised rows either (bit 7), at the same launch sizes: same g, the dh pass within
        # one rounding of dh * rstd, and bit-identical from run to run
        L.lib().vsx_set_flag(b"mlp_fused", 255)
        want = dh4.float() * r2[:, None]
        first = None
        for it in range(3):
            q7, cs2 = torch.zeros((B, H4), device="cuda"), torch.zeros((2, H4), device="cuda")
            (y7, mean7), r7, _, g7 = ops.mlp_fc1_ln(y, img, b1, q7, M, C, hw, 1e-6, store_h=False, store_xh=False)
            assert torch.equal(g2, g7) and torch.equal(r2, r7)
            dh7 = ops.mlp_bwd_dh_ln(dout, y, mean7, r7, img2, img, b1, s, t, cs2, M, C, hw)
            badrow = ((dh7.float() - want).abs() > 2.0 ** -7 * want.abs() + 1e-30).any(1).nonzero().flatten()
            assert badrow.numel() == 0, (it, badrow.numel(), badrow[:8].tolist())
            if first is None:
                first = dh7
            else:
                assert torch.equal(first, dh7)
            del g7, dh7
    finally:
        L.lib().vsx_set_flag(b"mlp_fused", saved)


def test_weight_task_list_equals_single_launches():
    """vsx_weight_tasks (ops.batch): prep_weight / transpose_f32 / matvec / mlp_pack / unprep_grad / matvec_t_add collected into task lists give identical
    outputs to the single launches — more tasks than one launch holds (VSX_WTASK_MAX = 40), every kind, bf16 and fp32, tap
    reordering, gamma fold, accumulate"""
    O = _hip()

    def jobs():
        out = []
        for i in range(30):
            Rr, Cs, Tn = 32 + 8 * (i % 5), 16 + 8 * (i % 3), (1, 4, 9, 27)[i % 4]
            src = torch.randn((Rr, Cs, Tn), generator=torch.Generator().manual_seed(100 + i)).cuda()
            gam = torch.randn(Cs, generator=torch.Generator().manual_seed(200 + i)).cuda() if i % 2 else None
            dt = torch.bfloat16 if i % 3 else torch.float32
            out.append(O.prep_weight(src, Rr, Cs, Tn, dt, want=i % 4 != 1, want_t=i % 4 != 2, gamma=gam, tapmode=1 if Tn == 27 else 0))
        for i in range(25):
            A, Bn = 7 + 13 * i, 49 if i % 2 else 1
            src = torch.randn((A, Bn), generator=torch.Generator().manual_seed(300 + i)).cuda()
            dst = torch.randn((Bn, A), generator=torch.Generator().manual_seed(400 + i)).cuda()
            O.transpose_f32(src, dst, A, Bn, bool(i % 2))
            out.append((dst,))
        for i in range(20):
            Rr, Cc = 24 + 40 * i, 96 + 32 * (i % 4)
            Wm = torch.randn((Rr, Cc), generator=torch.Generator().manual_seed(500 + i)).cuda()
            v = torch.randn(Cc, generator=torch.Generator().manual_seed(600 + i)).cuda()
            b = torch.randn(Rr, generator=torch.Generator().manual_seed(700 + i)).cuda() if i % 2 else None
            out.append((O.matvec(Wm, v, b, Rr, Cc),))
        for Cw in (96, 192, 224, 384):
            W1 = torch.randn((4 * Cw, Cw), generator=torch.Generator().manual_seed(800 + Cw)).cuda().bfloat16()
            W2 = torch.randn((Cw, 4 * Cw), generator=torch.Generator().manual_seed(900 + Cw)).cuda().bfloat16()
            out.append((O.mlp_pack(W1, W2, Cw),))
        for i in range(12):  # gradient finalisers: unprep_grad (plain / LayerNorm-affine unfold / tap reordering), matvec_t_add
            Rr, Cs, Tn = 64 + 32 * i, 24 + 8 * (i % 4), (1, 1, 4, 27)[i % 4]
            gs = lambda k, *sh: torch.randn(sh, generator=torch.Generator().manual_seed(1000 + 10 * i + k)).cuda()
            gW, dparam = gs(0, Rr, Cs * Tn), gs(1, Rr, Cs, Tn)
            aff = i % 4 == 1
            gam, Wp, dgam, u, beta = (gs(2, Cs), gs(3, Rr, Cs, Tn), gs(4, Cs), gs(5, Rr), gs(6, Cs)) if aff else (None,) * 5
            O.unprep_grad(gW, dparam, Rr, Cs, Tn, gamma=gam, W=Wp, dgamma=dgam, u=u, beta=beta, tapmode=1 if Tn == 27 else 0)
            acc = gs(7, Cs * Tn)
            O.matvec_t_add(gW, gs(8, Rr), acc, Rr, Cs * Tn)
            out.append((dparam, dgam, acc))
        return out

    single = jobs()
    with O.batch():
        listed = jobs()
    torch.cuda.synchronize()
    assert len(single) == len(listed) == 91
    for a, b in zip(single, listed):
        for ta, tb in zip(a, b):
            assert (ta is None) == (tb is None)
            if ta is not None:  # dgamma / matvec_t accumulate through atomics: order-dependent round-off
                assert torch.equal(ta, tb) or torch.allclose(ta, tb, rtol=1e-5, atol=1e-5)
