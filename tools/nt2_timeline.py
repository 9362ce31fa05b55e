"""GPU-only, probe build (python -m viscy_amd.build ts -DNT2_TS=1; VSX_LIB=viscy_amd/libvsx_ts.so): where a tile of the fused fc1
data gradient + LayerNorm backward (gemm_nt2_lnbwd_kernel) spends its cycles — s_memtime stamps of one workgroup's eight waves."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

dt = torch.bfloat16
for (M, Cc) in [(2097152, 224), (524288, 192), (2097152, 96)]:
    K = 4 * Cc
    dh = torch.randn(M, K, device="cuda").to(dt)
    W = (torch.randn(Cc, K, device="cuda") * K ** -0.5).to(dt)
    y = torch.randn(M, Cc, device="cuda").to(dt)
    rstd = torch.rand(M, device="cuda") + 0.5
    mean = torch.randn(M, device="cuda") * 0.1
    for _ in range(3):
        ops.dgrad_ln_bwd(dh, W, y, rstd, M, Cc, K, mean=mean)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 * 80))()
    L.lib().vsx_debug_nt2_ts.argtypes = [C.c_void_p]
    assert L.lib().vsx_debug_nt2_ts(buf) == 0
    nk = K // 32
    print(f"=== M={M} C={Cc} K={K} ({nk} slabs)")
    for w in range(8):
        t = [buf[w * 80 + i] for i in range(80)]
        slabs = [t[8 + i] for i in range(min(nk, 64))]
        d = [slabs[i + 1] - slabs[i] for i in range(len(slabs) - 1)]
        print(f"wave {w}: entry->slab0 {slabs[0] - t[0]:6d} | K loop {t[1] - slabs[0]:7d} (per slab min {min(d)} med {sorted(d)[len(d) // 2]} max {max(d)}) | "
              f"barrier {t[2] - t[1]:5d} | pass0 {t[3] - t[2]:6d} | pass1 {t[4] - t[3]:6d} | tail {t[5] - t[4]:6d} | TOTAL {t[5] - t[0]:7d}")
