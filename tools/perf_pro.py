"""GPU-only: the fc2 forward with the GRN prologue (stage 2 / 3 shapes of the bench step) on the fragment form (nt2 bit 4) and on
the LDS-side form one slab ahead (default), next to the plain launch of the same shape.  us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

dt = torch.bfloat16


def timeit(fn, n=40):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, hw) in [(131072, 384, 1536, 256), (32768, 768, 3072, 64), (2097152, 224, 896, 4096)]:
    A = torch.randn(M, K, device="cuda").to(dt)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    s = 1 + 0.2 * torch.randn(M // hw, K, device="cuda")
    beta = 0.1 * torch.randn(K, device="cuda")
    res = torch.randn(M, N, device="cuda").to(dt)
    bias = torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda", dtype=dt)
    r = {}

    def pro():
        ops.gemm("nt", A, W, C, M, N, K, K, K, N, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=beta, hw=hw, epi=L.EPI_BIAS_RES, bias=bias, res=res, ldr=N)

    def plain():
        ops.gemm("nt", A, W, C, M, N, K, K, K, N, dtype=dt, hw=hw, epi=L.EPI_BIAS_RES, bias=bias, res=res, ldr=N)

    base = L.lib().vsx_get_flag(b"nt2")
    for rep in range(2):
      for name, flag in (("lds-side", base & ~16), ("fragment", base | 16), ("lds-side, every launch", (base & ~16) | 2), ("fragment, every launch", base | 16 | 2)):
        L.lib().vsx_set_flag(b"nt2", flag)
        r[f"{name} #{rep}"] = (timeit(pro), L.lib().vsx_last_kernel().decode())
      L.lib().vsx_set_flag(b"nt2", base)
      r[f"plain #{rep}"] = (timeit(plain), L.lib().vsx_last_kernel().decode())
    print(f"M={M} N={N} K={K} hw={hw}:\n  " + "\n  ".join(f"{k} {v[0]:7.1f} us ({v[1]})" for k, v in r.items()), flush=True)
