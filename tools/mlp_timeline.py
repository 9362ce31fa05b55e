"""Probe builds only (csrc/mlp.hip compiled with -DMLP_TS=1): the s_memtime stamps of one workgroup's 8 waves at the phase
boundaries of every sub-chunk step of a dh pass — where the cycles of a step go.
    VSX_LIB=<probe .so> python tools/mlp_timeline.py [C] [mode: 4 5 7]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 224
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 7
B, hw = 512, {96: 4096, 192: 1024, 224: 4096, 384: 256}[Cc]
dt = torch.bfloat16
M, H4 = B * hw, 4 * Cc
y = (torch.randn(M, Cc, device="cuda") * 2).to(dt)
W1 = (torch.randn(H4, Cc, device="cuda") * Cc ** -0.5).to(dt)
W2 = (torch.randn(Cc, H4, device="cuda") * H4 ** -0.5).to(dt)
b1 = torch.randn(H4, device="cuda") * 0.1
colsq = torch.zeros((B, H4), device="cuda")
img, img2 = ops.mlp_pack(W1, W2, Cc), ops.mlp_pack(W2.t().contiguous(), W2, Cc)
s = 1 + 0.2 * torch.randn(B, H4, device="cuda")
t = 0.05 * torch.randn(B, H4, device="cuda")
dout = torch.randn(M, Cc, device="cuda").to(dt)
if mode in (0, 2, 6):
    fn = {0: lambda: ops.mlp_stats(y, img, b1, colsq, M, Cc, hw, ln_eps=1e-6),
          2: lambda: ops.mlp_fc1_ln(y, img, b1, colsq, M, Cc, hw, 1e-6),
          6: lambda: ops.mlp_fc1_ln(y, img, b1, colsq, M, Cc, hw, 1e-6, store_h=False, store_xh=False)}[mode]
elif mode == 3:
    xh, rstd, h, g = ops.mlp_fc1_ln(y, img, b1, colsq, M, Cc, hw, 1e-6)
    P, S = torch.zeros(B, H4, device="cuda"), torch.zeros(B, H4, device="cuda")
    fn = lambda: ops.mlp_bwd_stats(dout, img2, g, P, S, M, Cc, hw)  # noqa: E731
elif mode == 7:
    (_, mean), rstd, _, _ = ops.mlp_fc1_ln(y, img, b1, colsq, M, Cc, hw, 1e-6, store_h=False, store_xh=False)
    cs2 = torch.zeros((2, H4), device="cuda")
    fn = lambda: ops.mlp_bwd_dh_ln(dout, y, mean, rstd, img2, img, b1, s, t, cs2, M, Cc, hw)  # noqa: E731
else:
    xh, rstd, h, g = ops.mlp_fc1_ln(y, img, b1, colsq, M, Cc, hw, 1e-6)
    db = torch.zeros(H4, device="cuda")
    fn = (lambda: ops.mlp_bwd_dh(dout, img2, h, s, t, db, M, Cc, hw)) if mode == 4 else (lambda: ops.mlp_bwd_dh_re(dout, xh, img2, img, b1, s, t, db, M, Cc, hw))
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 64 * 8))()
L.lib().vsx_debug_mlp_ts.argtypes = [C.c_void_p]
assert L.lib().vsx_debug_mlp_ts(buf) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 8).astype(np.int64)
nhs = H4 // 32
names = ["barrier->start", "flush / tile hand-over (0->1)", "DMA issue + partial sums (1->2)", "next sub-chunk GEMM (2->3)", "activation (3->4)", "column sums / statistics (4->5)", "accumulator copy + vmcnt(0) (5->6)", "barrier wait (6->next 0)"]
for w in range(8):
    d = []
    for hsx in range(2, nhs - 2):
        r = ts[w, hsx]
        nxt0 = ts[w, hsx + 1, 0]
        d.append([r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], nxt0 - r[6], nxt0 - r[0]])
    d = np.array(d, dtype=np.float64)
    print(f"wave {w}: " + " | ".join(f"{n.split(' (')[0]} {v:7.0f}" for n, v in zip(names[1:] + ["STEP"], d.mean(0))))
ev = [ts[w, 4:nhs - 4, 0] for w in range(8)]
print("step start skew across waves (cycles, mean of |t_w - t_0|):", [int(np.abs(e - ev[0]).mean()) for e in ev])
print("clock: s_memtime ticks; 100 MHz constant clock? compare STEP with kernel time / steps")
