import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unext2_ref
from viscy_amd.unext2 import UNeXt2
kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True)
torch.manual_seed(0)
ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=7).eval()
mine = UNeXt2(**kw); mine.load_state_dict(ref.state_dict()); mine = mine.cuda(); mine.compute_dtype = torch.float32
x = torch.randn(2, 1, 5, 128, 128)
y = ref(x); dout = torch.randn_like(y); y.backward(dout)
for trial in range(2):
    for p in mine.parameters(): p.grad = None
    out = mine(x.cuda()); out.backward(dout.cuda())
    errs = []
    for (name, pr), pm in zip(ref.named_parameters(), mine.parameters()):
        e = ((pm.grad.cpu() - pr.grad).abs().max() / pr.grad.abs().max().clamp_min(1e-12)).item()
        errs.append((e, name))
    errs.sort(reverse=True)
    print("trial", trial, "fwd err", ((out.cpu()-y).abs().max()/y.abs().max()).item())
    for e, n in errs[:12]: print(f"  {e:.3e} {n}")
