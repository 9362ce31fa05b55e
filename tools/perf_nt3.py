"""Two launches for PMC analysis: d2 fc1-like with K=224 and K=32 (epilogue only), EPI_NONE."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import ops  # noqa: E402
from viscy_amd._lib import lib  # noqa: E402

lib().vsx_set_flag(b"nt_wide", 0)
dt, dev = torch.bfloat16, "cuda"
M, C, N4 = 524288, 224, 896
x, W1 = torch.randn(M, C, device=dev).to(dt), torch.randn(N4, C, device=dev).to(dt)
h = torch.empty(M, N4, device=dev, dtype=dt)
for K in (224, 32, 224, 32):
    ops.gemm("nt", x, W1, h, M, N4, K, C, C, N4, dtype=dt)
    torch.cuda.synchronize()
h2 = torch.empty(M, C, device=dev, dtype=dt)
W2 = torch.randn(C, N4, device=dev).to(dt)
for _ in range(2):
    ops.gemm("nt", h, W2, h2, M, C, N4, N4, N4, C, dtype=dt)
    torch.cuda.synchronize()
