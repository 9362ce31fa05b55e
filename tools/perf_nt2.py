"""Is the NT epilogue limited by the scattered 256-B-segment write pattern?  Same bytes, different output geometry."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402,F401
from viscy_amd import ops  # noqa: E402

dt, dev = torch.bfloat16, "cuda"


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


def rnd(*s):
    return torch.randn(*s, device=dev).to(dt)


M0 = 524288
for N, mult in [(896, 1), (128, 7), (256, 7), (512, 7), (1792, 1), (64, 14)]:
    M = M0 * mult if N < 896 else M0
    if N == 256:
        M = M0 * 7 // 2
    if N == 512:
        M = M0 * 7 // 4
    if N == 1792:
        M = M0 // 2
    K = 32
    x, W = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, device=dev, dtype=dt)
    us = timeit(lambda: ops.gemm("nt", x, W, out, M, N, K, K, K, N, dtype=dt))
    print(f"M={M:8d} N={N:5d} K=32 ldc=N : {us:8.1f} us  write {M * N * 2 / us / 1e3:7.1f} GB/s  ({M * N * 2 / 1e6:.0f} MB)")
