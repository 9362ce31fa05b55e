"""NT GEMM anatomy at the d2 / s0 shapes: full K vs K=32 (one k-step: epilogue + launch only), by epilogue kind."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

dt, dev = torch.bfloat16, "cuda"
from viscy_amd._lib import lib  # noqa: E402
lib().vsx_set_flag(b"nt_wide", int(os.environ.get("NTW", 1)))
lib().vsx_set_flag(b"tn_rect", int(os.environ.get("TNR", 1)))
B = int(os.environ.get("B", 128))


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


for name, hw, C in [("d2", 64, 224), ("s0", 64, 96), ("s1", 32, 192), ("s2", 16, 384)]:
    M, N4 = B * hw * hw, 4 * C
    x, W1 = rnd(M, C), rnd(N4, C)
    h, g = torch.empty(M, N4, device=dev, dtype=dt), torch.empty(M, N4, device=dev, dtype=dt)
    b1, colsq = torch.zeros(N4, device=dev), torch.zeros(B, N4, device=dev)
    for K in (C, 32):
        for epi, nm in [(L.EPI_NONE, "none"), (L.EPI_BIAS, "bias"), (L.EPI_BIAS_GELU_SQ, "gelu_sq+C2")]:
            kw = dict(epi=epi, bias=b1, red0=colsq, hw=hw * hw)
            if epi == L.EPI_BIAS_GELU_SQ:
                kw["C2"] = g
            us = timeit(lambda: ops.gemm("nt", x, W1, h, M, N4, K, C, C, N4, dtype=dt, **kw))
            wr = M * N4 * 2 * (2 if epi == L.EPI_BIAS_GELU_SQ else 1)
            print(f"{name} fc1-like M={M} N={N4} K={K:4d} epi={nm:11s}: {us:8.1f} us   write {wr / us / 1e3:7.1f} GB/s  total {(wr + M * K * 2) / us / 1e3:7.1f} GB/s")
    W2, out, res = rnd(C, N4), torch.empty(M, C, device=dev, dtype=dt), rnd(M, C)
    b2 = torch.zeros(C, device=dev)
    s, gb = torch.ones(B, N4, device=dev), torch.zeros(N4, device=dev)
    for K in (N4, 32):
        us = timeit(lambda: ops.gemm("nt", h, W2, out, M, C, K, N4, N4, C, dtype=dt))
        print(f"{name} fc2-like M={M} N={C} K={K:4d} epi=none       : {us:8.1f} us   read {M * K * 2 / us / 1e3:7.1f} GB/s")
    us = timeit(lambda: ops.gemm("nt", h, W2, out, M, C, N4, N4, N4, C, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=gb, hw=hw * hw, epi=L.EPI_BIAS_RES, bias=b2, res=res, ldr=C))
    print(f"{name} fc2 grn-prologue+bias_res                : {us:8.1f} us   read {M * N4 * 2 / us / 1e3:7.1f} GB/s")
    dW = torch.zeros(N4, C, device=dev)
    us = timeit(lambda: ops.gemm("tn", x, h, dW, M, N4, C, C, N4, C, dtype=dt))
    print(f"{name} tn dW1 M={M} N={N4} K={C}: {us:8.1f} us  {(M * C + M * N4) * 2 / us / 1e3:8.1f} GB/s")
    dW2 = torch.zeros(C, N4, device=dev)
    us = timeit(lambda: ops.gemm("tn", h, out, dW2, M, C, N4, N4, C, N4, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=gb, hw=hw * hw))
    print(f"{name} tn dW2[grn] M={M} N={C} K={N4}: {us:8.1f} us  {(M * C + M * N4) * 2 / us / 1e3:8.1f} GB/s")
    # reference points: pure streaming of the same bytes
    us = timeit(lambda: torch.add(h, 1.0, out=g))
    print(f"{name} torch.add {M}x{N4} bf16 (read+write): {us:8.1f} us  {2 * M * N4 * 2 / us / 1e3:8.1f} GB/s")
    us = timeit(lambda: g.copy_(h))
    print(f"{name} copy_ {M}x{N4} bf16: {us:8.1f} us  {2 * M * N4 * 2 / us / 1e3:8.1f} GB/s")
    us = timeit(lambda: h.zero_())
    print(f"{name} zero_ {M}x{N4} bf16 (write only): {us:8.1f} us  {M * N4 * 2 / us / 1e3:8.1f} GB/s")
