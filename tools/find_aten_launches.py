"""Which ATen / runtime launches (copies, fills, elementwise kernels) are left inside the training step?  One eager step of the bench
configuration under torch.profiler with Python stacks: every device-side activity that is not a libvsx kernel, by call site.
(the captured step should hold hand-written kernels only.)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from viscy_amd.losses import MixedLoss  # noqa: E402
from viscy_amd.optim import FlatAdamW  # noqa: E402
from viscy_amd.parallel import FlatDataParallel  # noqa: E402
from viscy_amd.step import TrainStep  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4,
               decoder_conv_blocks=2).to(dev)
bench.nonzero_grn_(model)
model.compute_dtype, model.grad_mode = torch.bfloat16, "flat"
eng = model.engine()
opt = FlatAdamW(eng, lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=10, warmup_multiplier=1e-3)
ddp = FlatDataParallel(eng, opt)
x, tgt = bench.make_batch(B, 256, 256, dev)
step = TrainStep(model, MixedLoss(0.5, 0.0, 0.5), opt, ddp, use_graph=False, static_inputs=True)
for _ in range(3):
    step(x, tgt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x, tgt)
    torch.cuda.synchronize()
kernels = collections.Counter()
for ev in prof.events():
    if str(getattr(ev, "device_type", "")).endswith("CUDA"):
        kernels[ev.name[:90]] += 1
ours = ("mlp_", "gemm_", "dwconv", "head_", "ln_", "grn_", "ssim_", "loss_", "weight_tasks", "adamw", "ps_cat", "stem_", "reduce_rows",
        "fill_f32", "tn_zero", "scale_", "prep_", "pad_cols", "normalize", "dw_reduce", "transpose", "matvec", "unprep", "layer_scale")
foreign = {k: v for k, v in kernels.items() if not any(o in k for o in ours)}
print(f"{sum(kernels.values())} device activities in one eager step (B = {B}); not libvsx kernels: {sum(foreign.values())}")
for k, v in sorted(foreign.items(), key=lambda kv: -kv[1]):
    print(f"  {v:5d} x {k}")
sites = collections.Counter()
launching = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::div", "aten::clone", "aten::to",
             "aten::_to_copy", "aten::lt", "aten::rand", "aten::uniform_", "aten::cat", "aten::sum", "aten::sqrt", "aten::where", "aten::index",
             "aten::masked_fill_", "aten::sub", "aten::neg", "aten::contiguous", "aten::zeros", "aten::ones", "aten::full", "aten::zeros_like")
for ev in prof.events():
    if ev.name in launching and not str(getattr(ev, "device_type", "")).endswith("CUDA"):
        here = [f for f in (ev.stack or []) if "/viscy_amd/" in f or "bench.py" in f]
        sites[(ev.name, here[0].strip() if here else "?")] += 1
print("ATen calls that may launch, by call site:")
for (name, where), n in sites.most_common(50):
    print(f"{n:5d} x {name:22s} {where}")
