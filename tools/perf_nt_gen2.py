"""The step's non-prologue NT launches (B = 512) on the first- and second-generation kernels: us, GB/s of algorithmic bytes, TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

dt, dev = torch.bfloat16, "cuda"
l = L.lib()
B = int(os.environ.get("B", 512))


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


def rnd(*s):
    return torch.randn(*s, device=dev).to(dt)


# (name, hw, M-rows per sample side, N, K, epi)
SHAPES = []
for name, side, C in [("s0", 64, 96), ("s1", 32, 192), ("s2", 16, 384), ("s3", 8, 768), ("d2", 64, 224)]:
    SHAPES += [(f"{name} fc1 e2", side, 4 * C, C, L.EPI_BIAS_GELU_SQ), (f"{name} dz e4", side, 4 * C, C, L.EPI_DZ),
               (f"{name} fc2 e3", side, C, 4 * C, L.EPI_BIAS_RES), (f"{name} dgrad e0", side, C, 4 * C, L.EPI_NONE),
               (f"{name} fc2 e3 grn", side, C, 4 * C, -L.EPI_BIAS_RES)]
only = os.environ.get("ONLY")
flags = [int(f) for f in os.environ.get("NT2", "0,3").split(",")]
for name, side, N, K, epi in SHAPES:
    if only and only not in name:
        continue
    hw = side * side
    M = B * hw
    A, W = rnd(M, K), rnd(N, K) * K**-0.5
    C1, C2 = torch.empty(M, N, device=dev, dtype=dt), torch.empty(M, N, device=dev, dtype=dt)
    bias, r0, r1 = torch.zeros(N, device=dev), torch.zeros(B, N, device=dev), torch.zeros(B, N, device=dev)
    grn = epi < 0
    epi = abs(epi)
    kw = dict(dtype=dt, hw=hw, epi=epi)
    if grn:
        kw.update(pro=L.PRO_GRN, grn_s=torch.ones(B, K, device=dev), grn_b=torch.zeros(K, device=dev))
    nbytes = (M * K + M * N + N * K) * 2
    if epi == L.EPI_BIAS_GELU_SQ:
        kw.update(bias=bias, red0=r0, C2=C2)
        nbytes += M * N * 2
    elif epi == L.EPI_DZ:
        kw.update(aux=C2, ldx=N, red0=r0, red1=r1)
        nbytes += M * N * 2
    elif epi == L.EPI_BIAS_RES:
        kw.update(bias=bias, res=C2, ldr=N)
        nbytes += M * N * 2
    row = f"{name:14s} M={M:8d} N={N:5d} K={K:5d}"
    for f in flags:
        l.vsx_set_flag(b"nt2", f)
        us = timeit(lambda: ops.gemm("nt", A, W, C1, M, N, K, K, K, N, **kw))
        row += f" | nt2={f}: {us:8.1f} us {nbytes / us / 1e3:7.0f} GB/s {2.0 * M * N * K / us / 1e6:6.0f} TF"
    print(row, flush=True)
    del A, W, C1, C2
