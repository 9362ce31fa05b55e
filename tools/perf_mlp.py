"""GPU-only: fused GRN-MLP (csrc/mlp.hip) vs the unfused fc1 / fc2 launches, per block shape of the tiny backbone at a
given batch (default 512 patches of 256x256: the bench workload).  Prints us per launch, TFLOP/s and algorithmic GB/s."""
import sys

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SHAPES = [(96, 4096, 3), (192, 1024, 5), (224, 4096, 2)]  # C, hw per patch, blocks per forward (C = 384: training passes only, tools/perf_mlp_train.py)
L.lib().vsx_set_flag(b"mlp_fused", 255)
dt = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {"fused_inf": 0.0, "unfused_inf": 0.0}
for C, hw, nblk in SHAPES:
    M, H4 = B * hw, 4 * C
    xh = torch.randn(M, C, device="cuda").to(dt)
    res = torch.randn(M, C, device="cuda").to(dt)
    W1 = (torch.randn(H4, C, device="cuda") * C ** -0.5).to(dt)
    W2 = (torch.randn(C, H4, device="cuda") * H4 ** -0.5).to(dt)
    b1, b2 = torch.randn(H4, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    gamma, beta = torch.randn(H4, device="cuda") * 0.3, torch.randn(H4, device="cuda") * 0.1
    colsq = torch.zeros((B, H4), device="cuda")
    img = ops.mlp_pack(W1, W2, C)
    ops.mlp_stats(xh, img, b1, colsq, M, C, hw)
    s = ops.grn_scale(colsq, gamma)
    out = torch.empty((M, C), dtype=dt, device="cuda")
    g = torch.empty((M, H4), dtype=dt, device="cuda")
    h = torch.empty((M, H4), dtype=dt, device="cuda")
    f1 = 2.0 * M * H4 * C
    t_stats = timeit(lambda: ops.mlp_stats(xh, img, b1, colsq, M, C, hw))
    t_out = timeit(lambda: ops.mlp_out(xh, img, b1, s, beta, b2, res, None, M, C, hw))
    t_fc1_inf = timeit(lambda: ops.gemm("nt", xh, W1, None, M, H4, C, C, C, H4, dtype=dt, epi=L.EPI_BIAS_GELU_SQ, bias=b1, red0=colsq, hw=hw, C2=g))
    t_fc1_tr = timeit(lambda: ops.gemm("nt", xh, W1, h, M, H4, C, C, C, H4, dtype=dt, epi=L.EPI_BIAS_GELU_SQ, bias=b1, red0=colsq, hw=hw, C2=g))
    t_fc1_fused = timeit(lambda: ops.mlp_fc1(xh, img, b1, colsq, M, C, hw), n=5)
    print(f"      fc1(train): unfused GEMM {t_fc1_tr:8.1f} us | fused kernel + h/g stores {t_fc1_fused:8.1f} us ({2 * M * H4 * 2 / t_fc1_fused / 1e3:6.0f} GB/s stored)")
    # block backward: unfused dz GEMM (EPI_DZ) + grn_gelu_bwd vs the two recompute passes (MODE 3 / 4)
    dout = torch.randn(M, C, device="cuda").to(dt)
    W2T = W2.t().contiguous()
    tt = torch.randn(B, H4, device="cuda") * 0.05
    PS = torch.zeros((2, B, H4), device="cuda")
    dzb = torch.empty((M, H4), dtype=dt, device="cuda")
    dbb = torch.zeros(H4, device="cuda")
    img2 = ops.mlp_pack(W2T, W2, C)
    t_dz = timeit(lambda: ops.gemm("nt", dout, W2T, dzb, M, H4, C, C, C, H4, dtype=dt, epi=L.EPI_DZ, aux=g, ldx=H4, red0=PS[0], red1=PS[1], hw=hw), n=5)
    t_ggb = timeit(lambda: ops.grn_gelu_bwd(dzb, h, s, tt, dbb, M, H4, hw), n=5)
    t_b3 = timeit(lambda: ops.mlp_bwd_stats(dout, img2, g, PS[0], PS[1], M, C, hw), n=5)
    t_b4 = timeit(lambda: ops.mlp_bwd_dh(dout, img2, h, s, tt, dbb, M, C, hw), n=5)
    Qb = torch.empty((B, C, H4), device="cuda")
    csb = torch.empty((B, C), device="cuda")
    dW2 = torch.zeros((C, H4), device="cuda")
    db2 = torch.zeros(C, device="cuda")
    t_tn_pro = timeit(lambda: ops.gemm("tn", g, dout, dW2, M, C, H4, H4, C, H4, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=beta, hw=hw, colsum=db2), n=5)
    t_tn_ps = timeit(lambda: ops.gemm("tn", g, dout, Qb, M, C, H4, H4, C, H4, dtype=dt, hw=hw, colsum=csb, b_bstride=C * H4), n=5)
    t_qr = timeit(lambda: ops.grn_q_reduce(Qb, csb, W2, s, beta, PS[0], PS[1], dW2, db2), n=5)
    print(f"      dW2: TN with GRN prologue {t_tn_pro:8.1f} us | per-sample TN {t_tn_ps:8.1f} + q_reduce {t_qr:8.1f} us (also yields P, S)")
    del Qb
    print(f"      backward: dz GEMM {t_dz:8.1f} + grn_gelu_bwd {t_ggb:8.1f} = {t_dz + t_ggb:8.1f} us | recompute passes: stats {t_b3:8.1f} + dh {t_b4:8.1f} = {t_b3 + t_b4:8.1f} us")
    del dzb
    t_fc2 = timeit(lambda: ops.gemm("nt", g, W2, out, M, C, H4, H4, H4, C, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=beta, hw=hw, epi=L.EPI_BIAS_RES, bias=b2, res=res, ldr=C))
    Ws = ops.scale_weight_samples(W2.float(), s, dt) if hw % 128 == 0 and hw // 128 >= 8 else None
    t_fc2f = timeit(lambda: ops.gemm("nt", g, Ws, out, M, C, H4, H4, H4, C, dtype=dt, hw=hw, b_bstride=C * H4, epi=L.EPI_BIAS_RES, bias=b2, res=res, ldr=C)) if Ws is not None else float("nan")
    best_fc2 = min(t_fc2, t_fc2f) if Ws is not None else t_fc2
    print(f"C={C:4d} hw={hw:5d} M={M:8d}: stats {t_stats:8.1f} us ({f1 / t_stats / 1e6:6.0f} TF/s) | out {t_out:8.1f} us ({2 * f1 / t_out / 1e6:6.0f} TF/s, "
          f"{3 * M * C * 2 / t_out / 1e3:6.0f} GB/s) || fc1(inf) {t_fc1_inf:8.1f} fc1(train) {t_fc1_tr:8.1f} fc2(pro) {t_fc2:8.1f} fc2(fold) {t_fc2f:8.1f} "
          f"|| fused {t_stats + t_out:8.1f} vs unfused {t_fc1_inf + best_fc2:8.1f} us")
    tot["fused_inf"] += nblk * (t_stats + t_out)
    tot["unfused_inf"] += nblk * (t_fc1_inf + best_fc2)
print({k: round(v / 1e3, 2) for k, v in tot.items()}, "ms per forward (MLP part, C=768 blocks excluded)")
