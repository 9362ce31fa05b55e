"""Repeat the bf16 engine's forward + MixedLoss + backward on fixed inputs and compare every repetition's gradients with the
first, per parameter group: atomics-order noise is ~1e-3 relative at most; a race shows as an outlier in one group.
env: N (repetitions), B, S (batch / patch size), VSX_FLAGS for bisection."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd.losses import MixedLoss  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

N = int(os.environ.get("N", 40))
B = int(os.environ.get("B", 4))
S = int(os.environ.get("S", 256))
torch.manual_seed(13)
m = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4,
           decoder_conv_blocks=2).cuda()
with torch.no_grad():
    for n, p in m.named_parameters():
        if "grn" in n:
            p.normal_(0, 0.2)
m.compute_dtype, m.grad_mode = torch.bfloat16, "flat"
eng = m.engine()
g = torch.Generator().manual_seed(2024)
x = torch.randn((B, 1, 5, S, S), generator=g).cuda()
t = torch.randn((B, 2, 5, S, S), generator=g).cuda()
loss_fn = MixedLoss(0.5, 0.0, 0.5)
named = dict(m.named_parameters())
groups = {}
for n in named:
    k = ".".join(n.split(".")[:3]) if n.startswith(("encoder_stages", "decoder")) else n.split(".")[0]
    groups.setdefault(k, []).append(n)
ref = None
ys = None
bad = 0
for it in range(N):
    eng.flat_grad.zero_()
    y = m(x)
    loss = loss_fn(y, t)
    loss.backward()
    torch.cuda.synchronize()
    cur = {n: eng.g(named[n]).detach().clone() for n in named}
    if ref is None:
        ref, ys = cur, y.detach().clone()
        continue
    dy = ((y.detach() - ys).abs().max() / ys.abs().max()).item()
    worst = []
    for k, names in groups.items():
        a = torch.cat([ref[n].flatten() for n in names]).double()
        b = torch.cat([cur[n].flatten() for n in names]).double()
        rel = ((a - b).norm() / a.norm()).item()
        worst.append((rel, k))
    worst.sort(reverse=True)
    flag = worst[0][0] > float(os.environ.get("TOL", 2e-2)) or dy > 3e-2
    bad += flag
    if flag or it < 3:
        print(f"it {it}: fwd diff {dy:.2e}; worst groups " + ", ".join(f"{k} {r:.2e}" for r, k in worst[:4]), flush=True)
print(f"{bad} outlier repetition(s) of {N - 1}")
