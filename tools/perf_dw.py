"""GPU-only: depthwise 7x7 forward / data gradient / weight gradient per block shape of the tiny backbone at B = 512 (default),
under each value of the `dw_mfma` flag given on the command line (default: 7 = register-staged tiles, 15 = LDS-DMA tiles).
Prints us per launch and algorithmic GB/s (forward: read x, write y; data gradient: + the shortcut operand)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

B = int(os.environ.get("B", 512))
flags = [int(a) for a in sys.argv[1:]] or [7, 15]
SHAPES = [(64, 64, 96), (32, 32, 192), (16, 16, 384), (64, 64, 224), (32, 32, 192)]
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


for H, W, C in SHAPES[:4]:
    M = B * H * W
    x = torch.randn(M, C, device="cuda").to(dt)
    dy = torch.randn(M, C, device="cuda").to(dt)
    add = torch.randn(M, C, device="cuda").to(dt)
    w = torch.randn(49, C, device="cuda") * 0.1
    b = torch.randn(C, device="cuda")
    gb = M * C * 2 / 1e3
    line = f"{H:3d}x{W:3d}x{C:4d}:"
    for f in flags:
        L.lib().vsx_set_flag(b"dw_mfma", f)
        tf = timeit(lambda: ops.dwconv7_fwd(x, w, b, B, H, W, C))
        td = timeit(lambda: ops.dwconv7_bwd_data(dy, w, add, B, H, W, C))
        dw, db = torch.zeros(49, C, device="cuda"), torch.zeros(C, device="cuda")
        tw = timeit(lambda: ops.dwconv7_bwd_weight(dy, x, dw, db, B, H, W, C))
        line += f" | flag {f:2d}: fwd {tf:7.1f} us {2 * gb / tf:5.0f} GB/s, dgrad {td:7.1f} us {3 * gb / td:5.0f} GB/s, wgrad {tw:7.1f} us {2 * gb / tw:5.0f} GB/s"
    print(line, flush=True)
