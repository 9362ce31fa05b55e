"""Run-to-run deviation of every parameter gradient in bf16 mode (identical inputs, two backward passes)."""
import sys

import torch

sys.path.insert(0, ".")
from viscy_amd.unext2 import UNeXt2  # noqa: E402

torch.manual_seed(3)
m = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True).cuda()
with torch.no_grad():
    for n, p in m.named_parameters():
        if ".grn." in n:
            p.normal_(0.0, 0.2)
m.compute_dtype = torch.bfloat16
m.grad_mode = "flat"
eng = m.engine()
g = torch.Generator().manual_seed(12)
x = torch.randn(2, 1, 5, 256, 256, generator=g).cuda()
dout = torch.randn(2, 2, 5, 256, 256, generator=g).cuda()
outs, grads = [], []
for it in range(3):
    eng.flat_grad.zero_()
    y = m(x)
    y.backward(dout)
    outs.append(y.detach().clone())
    grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
print("fwd dev", ((outs[0] - outs[1]).abs().max() / outs[0].abs().max()).item(), ((outs[0] - outs[2]).abs().max() / outs[0].abs().max()).item())
rows = []
for n in grads[0]:
    a, b = grads[0][n], grads[1][n]
    rows.append((((a - b).norm() / a.norm().clamp_min(1e-20)).item(), a.norm().item(), n))
rows.sort(reverse=True)
tot = sum(r[1] ** 2 for r in rows) ** 0.5
for r in rows[:14]:
    print(f"rel dev {r[0]:.3e}  |g| {r[1]:.3e} ({r[1] / tot:.3f} of total)  {r[2]}")
