"""grn_gelu_bwd launch-geometry sweep (vsx_set_flag "ggb_blocks") at the stage shapes of the B=512 step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import ops  # noqa: E402
from viscy_amd._lib import lib  # noqa: E402

B = int(os.environ.get("B", 512))
dev, dt = "cuda", torch.bfloat16


def timeit(fn, n=10):
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


for name, hw, C in [("d2", 64, 224), ("s0", 64, 96), ("s1", 32, 192), ("s2", 16, 384), ("s3", 8, 768)]:
    M, N4 = B * hw * hw, 4 * C
    dz, h = torch.randn(M, N4, device=dev).to(dt), torch.randn(M, N4, device=dev).to(dt)
    s, t, cs = torch.ones(B, N4, device=dev), torch.zeros(B, N4, device=dev), torch.zeros(N4, device=dev)
    row = []
    for blocks in (512, 1024, 1536, 1792, 2048):
        lib().vsx_set_flag(b"ggb_blocks", blocks)
        us = timeit(lambda: ops.grn_gelu_bwd(dz, h, s, t, cs, M, N4, hw * hw))
        row.append(f"{blocks}: {us:7.1f} us {3 * M * N4 * 2 / us / 1e3:6.0f} GB/s")
    print(name, f"M={M} N={N4} |", " | ".join(row), flush=True)
    del dz, h
