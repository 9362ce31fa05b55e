"""MixedLoss forward + backward at the bench shape (B x 2 x 5 x 256 x 256), event-timed, for each value of the loss_fused
flag.  Under `rocprofv3 --kernel-trace --stats` the per-kernel split comes out of the trace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd.losses import MixedLoss  # noqa: E402

B = int(os.environ.get("B", 512))
REP = int(os.environ.get("REP", 10))
shape = (B, 2, 5, 256, 256)
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.rand(shape, device="cuda", generator=g)
p0 = t + 0.1 * torch.randn(shape, device="cuda", generator=g)
fn = MixedLoss(0.5, 0.0, 0.5)
for flag in [int(v) for v in os.environ.get("FLAGS", "0,1").split(",")]:
    L.lib().vsx_set_flag(b"loss_fused", flag)
    vals = []
    for it in range(REP + 2):
        p = p0.clone().requires_grad_(True)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        loss = fn(p, t)
        e1.record()
        loss.backward()
        e2.record()
        torch.cuda.synchronize()
        if it >= 2:
            vals.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    f = sorted(v[0] for v in vals)[len(vals) // 2]
    b = sorted(v[1] for v in vals)[len(vals) // 2]
    print(f"loss_fused={flag}: forward {f:.3f} ms  backward {b:.3f} ms  total {f + b:.3f} ms  loss {loss.item():.6f}", flush=True)
