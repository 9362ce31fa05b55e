"""GPU-only: run-to-run determinism of the bf16 training-schedule forward (same weights, same input): max |y1 - y2| / max |y| and how
many outputs differ, per flag set.  A forward has no atomics in front of a rounding except the GRN / InstanceNorm statistics (fp32
sums whose order varies): differences beyond a few bf16 ulps on isolated outputs mean a race."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
m = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True).cuda()
with torch.no_grad():
    for n, p in m.named_parameters():
        if ".grn." in n:
            p.normal_(0, 0.1)
m.compute_dtype, m.grad_mode = torch.bfloat16, "flat"
eng = m.engine()
x = torch.randn((B, 1, 5, S, S), device="cuda")
def timed_forward():
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y, sv = eng.forward(x, torch.bfloat16, need_bwd=True)
    e1.record()
    torch.cuda.synchronize()
    eng._pending_bwd = 0
    return y.clone(), e0.elapsed_time(e1)


# round 5: fp32 atomics (default) against vsx_set_flag("det_reduce", 1) — fixed-order GRN / InstanceNorm sums — and what it costs
for flags in ([("det_reduce", 0)], [("det_reduce", 1)], [("det_reduce", 0), ("mlp_fused", 47)], [("det_reduce", 1), ("mlp_fused", 47)]):
    L.lib().vsx_set_flag(b"mlp_fused", 239)
    for k, v in flags:
        L.lib().vsx_set_flag(k.encode(), v)
    timed_forward()
    ys, ms = [], []
    for i in range(4):
        y, t = timed_forward()
        ys.append(y)
        ms.append(t)
    ref = ys[0]
    for i in range(1, 4):
        d = (ys[i] - ref).abs()
        print(flags, f"run {i}: max diff / max |y| = {(d.max() / ref.abs().max()).item():.3e}, differing {int((d > 0).sum())} of {d.numel()}, "
              f"> 1e-2 * max: {int((d > 1e-2 * ref.abs().max()).sum())}", flush=True)
    print(flags, f"forward (training schedule, eager) {sorted(ms)[len(ms) // 2]:.2f} ms", flush=True)
L.lib().vsx_set_flag(b"det_reduce", 0)
L.lib().vsx_set_flag(b"mlp_fused", 239)
