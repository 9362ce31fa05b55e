"""dw_mfma bit 5 (two pixels per LDS access in the LDS-DMA depthwise kernel) against bit 5 off: identical bits, several shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

dt = torch.bfloat16
old = L.lib().vsx_get_flag(b"dw_mfma")
ok = True
for B, H, W, C in [(2, 64, 64, 96), (3, 32, 32, 192), (2, 16, 16, 384), (1, 40, 72, 224), (1, 33, 19, 32), (2, 16, 24, 64), (5, 48, 17, 64)]:
    g = torch.Generator().manual_seed(1)
    M = B * H * W
    x = torch.randn(M, C, generator=g).to(dt).cuda()
    add = torch.randn(M, C, generator=g).to(dt).cuda()
    w = (torch.randn(49, C, generator=g) * 0.2).cuda()
    b = torch.randn(C, generator=g).cuda()
    res = {}
    for f in (31, 63):
        L.lib().vsx_set_flag(b"dw_mfma", f)
        res[f] = (ops.dwconv7_fwd(x, w, b, B, H, W, C).clone(), ops.dwconv7_bwd_data(x, w, add, B, H, W, C).clone(),
                  ops.dwconv7_bwd_data(x, w, None, B, H, W, C).clone())
    L.lib().vsx_set_flag(b"dw_mfma", old)
    same = all(torch.equal(a, c) for a, c in zip(res[31], res[63]))
    ok &= same
    print((B, H, W, C), "identical" if same else "DIFFERENT", [float((a.float() - c.float()).abs().max()) for a, c in zip(res[31], res[63])], flush=True)
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
