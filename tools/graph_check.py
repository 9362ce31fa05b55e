import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd.losses import MixedLoss
from viscy_amd.optim import FlatAdamW
from viscy_amd.step import TrainStep
from viscy_amd.unext2 import UNeXt2
kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_femto", head_pool=True)
g = torch.Generator().manual_seed(3)
x = torch.randn((2, 1, 5, 192, 192), generator=g).cuda(); t = torch.rand((2, 2, 5, 192, 192), generator=g).cuda()
for mode in ("eager", "eager", "graph"):
    torch.manual_seed(0)
    m = UNeXt2(**kw).cuda(); m.compute_dtype, m.grad_mode = torch.float32, "flat"
    opt = FlatAdamW(m.engine(), lr=1e-3)
    step = TrainStep(m, MixedLoss(0.5, 0, 0.5), opt, use_graph=(mode == "graph"))
    print(mode, [round(float(step(x, t)), 5) for _ in range(7)], "psum", float(m.engine().flat.double().sum()))
