"""GPU-only: the training passes of the fused GRN-MLP family (csrc/mlp.hip) one by one, per block shape of the tiny backbone at the
bench batch: statistics only (MODE 0), fc1 storing h + g (MODE 2) / g only (MODE 6), dh pass reading h (MODE 4) / recomputing it
(MODE 5).  us per launch and design bytes / s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SHAPES = [(96, 4096), (192, 1024), (224, 4096)]
L.lib().vsx_set_flag(b"mlp_fused", 255)
dt = torch.bfloat16


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, hw in SHAPES:
    M, H4 = B * hw, 4 * C
    y = (torch.randn(M, C, device="cuda") * 2).to(dt)
    W1 = (torch.randn(H4, C, device="cuda") * C ** -0.5).to(dt)
    W2 = (torch.randn(C, H4, device="cuda") * H4 ** -0.5).to(dt)
    b1 = torch.randn(H4, device="cuda") * 0.1
    colsq = torch.zeros((B, H4), device="cuda")
    img, img2 = ops.mlp_pack(W1, W2, C), ops.mlp_pack(W2.t().contiguous(), W2, C)
    s = 1 + 0.2 * torch.randn(B, H4, device="cuda")
    t = 0.05 * torch.randn(B, H4, device="cuda")
    dout = torch.randn(M, C, device="cuda").to(dt)
    db = torch.zeros(H4, device="cuda")
    xh, rstd, h, g = ops.mlp_fc1_ln(y, img, b1, colsq, M, C, hw, 1e-6)
    cw, hwd = M * C * 2 / 1e3, M * H4 * 2 / 1e3  # KB-ish units -> GB/s below via us
    r = {}
    r["m0 stats"] = (timeit(lambda: ops.mlp_stats(y, img, b1, colsq, M, C, hw, ln_eps=1e-6)), cw)
    r["m2 h+g"] = (timeit(lambda: ops.mlp_fc1_ln(y, img, b1, colsq, M, C, hw, 1e-6)), 2 * cw + 2 * hwd)
    r["m6 g"] = (timeit(lambda: ops.mlp_fc1_ln(y, img, b1, colsq, M, C, hw, 1e-6, store_h=False)), 2 * cw + hwd)
    r["m4 dh(h)"] = (timeit(lambda: ops.mlp_bwd_dh(dout, img2, h, s, t, db, M, C, hw)), cw + 2 * hwd)
    r["m5 dh(re)"] = (timeit(lambda: ops.mlp_bwd_dh_re(dout, xh, img2, img, b1, s, t, db, M, C, hw)), 2 * cw + hwd)
    (_, mean), rstd, _, _ = ops.mlp_fc1_ln(y, img, b1, colsq, M, C, hw, 1e-6, store_h=False, store_xh=False)
    cs2 = torch.zeros((2, H4), device="cuda")
    r["m6 g, no x^"] = (timeit(lambda: ops.mlp_fc1_ln(y, img, b1, colsq, M, C, hw, 1e-6, store_h=False, store_xh=False)), cw + hwd)
    r["m7 dh(ln)"] = (timeit(lambda: ops.mlp_bwd_dh_ln(dout, y, mean, rstd, img2, img, b1, s, t, cs2, M, C, hw)), 2 * cw + hwd)
    print(f"C={C:4d} hw={hw:5d} M={M:8d}: " + " | ".join(f"{k} {us:7.1f} us {kb / us:6.0f} GB/s" for k, (us, kb) in r.items()), flush=True)
