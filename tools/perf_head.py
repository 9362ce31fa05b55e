"""GPU-only: the PixelToVoxelHead launches of the bench step one by one (B = 512, 64 x 64 x 224 decoder map -> 128 x 128 x 7 x 8 ->
conv 3x3x3 -> 5 x 32 -> IN / PReLU / 1x1x1 / shuffle -> (B, 2, 5, 256, 256)): us per launch and GB/s on the bytes each one must move."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import ops  # noqa: E402

from viscy_amd import _lib as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 512
for f in sys.argv[1:]:  # flag=value (e.g. head_rows=0: the thread-per-voxel passes)
    if "=" in f:
        k, v = f.split("=")
        assert L.lib().vsx_set_flag(k.encode(), int(v)) == 0, f
dt = torch.bfloat16
h = w = 64
C3, D, Zo, Cmid, Cout = 8, 7, 5, 32, 2
H2, W2 = 2 * h, 2 * w


def timeit(fn, n=8):
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


dec = torch.randn(B * h * w, 4 * C3 * D, device="cuda").to(dt)
hin = ops.head_shuffle_fwd(dec, B, h, w, C3, D, True)
Wc = (torch.randn(Cmid, 27 * C3, device="cuda") * 0.07).to(dt)
bias = torch.randn(Cmid, device="cuda") * 0.1
stats = torch.zeros(2, B, Cmid, device="cuda")
U = ops.head_conv_fwd(hin, Wc, bias, stats[0], stats[1], B, H2, W2, C3, Cmid, Zo)
w2 = torch.randn(4 * Cout, Cmid, device="cuda") * 0.2
b2 = torch.randn(4 * Cout, device="cuda") * 0.1
alpha = torch.full((1,), 0.25, device="cuda")
out = ops.head_out_fwd(U, stats[0], stats[1], w2, b2, alpha, B, H2, W2, Zo, Cmid, Cout)
dout = torch.randn_like(out)
S = torch.zeros(2, B, Cmid, device="cuda")
dalpha, dW2, db2 = torch.zeros(1, device="cuda"), torch.zeros(4 * Cout, Cmid, device="cuda"), torch.zeros(4 * Cout, device="cuda")
dv = ops.head_out_bwd1_wgrad(U, stats[0], stats[1], w2, alpha, dout, S[0], S[1], dalpha, dW2, db2, B, H2, W2, Zo, Cmid, Cout)
dU = ops.head_out_bwd2(U, stats[0], stats[1], w2, alpha, dv, S[0], S[1], B, H2, W2, Zo, Cmid, Cout)
dW, db = torch.zeros(Cmid, 27 * C3, device="cuda"), torch.zeros(Cmid, device="cuda")
Wp = ops.head_conv_dgrad_prep(Wc)
dhin = ops.head_conv_dgrad(dU, Wp, B, H2, W2, C3, Cmid, Zo)
gb = lambda *ts: sum(t.numel() * t.element_size() for t in ts) / 1e9
rows = [
    ("head_shuffle_fwd", lambda: ops.head_shuffle_fwd(dec, B, h, w, C3, D, True), gb(dec, hin)),
    ("head_conv_fwd", lambda: ops.head_conv_fwd(hin, Wc, bias, stats[0], stats[1], B, H2, W2, C3, Cmid, Zo), gb(hin, U)),
    ("head_out_fwd", lambda: ops.head_out_fwd(U, stats[0], stats[1], w2, b2, alpha, B, H2, W2, Zo, Cmid, Cout), gb(U, out)),
    ("head_out_bwd1_wgrad", lambda: ops.head_out_bwd1_wgrad(U, stats[0], stats[1], w2, alpha, dout, S[0], S[1], dalpha, dW2, db2, B, H2, W2, Zo, Cmid, Cout), gb(U, dout, dv)),
    ("head_out_bwd2", lambda: ops.head_out_bwd2(U, stats[0], stats[1], w2, alpha, dv, S[0], S[1], B, H2, W2, Zo, Cmid, Cout), gb(U, dv, dU)),
    ("head_conv_wgrad", lambda: ops.head_conv_wgrad(hin, dU, dW, db, B, H2, W2, C3, Cmid, Zo), gb(hin, dU)),
    ("head_conv_dgrad", lambda: ops.head_conv_dgrad(dU, Wp, B, H2, W2, C3, Cmid, Zo), gb(dU, dhin)),
    ("head_shuffle_bwd", lambda: ops.head_shuffle_bwd(dhin, B, h, w, C3, D, True), gb(dhin, dec)),
]
tot = 0.0
for name, fn, g in rows:
    us = timeit(fn)
    tot += us
    print(f"{name:22s} {us:8.1f} us  {g:5.2f} GB  {g / us * 1e6:6.0f} GB/s", flush=True)
print(f"sum {tot / 1e3:.2f} ms")
