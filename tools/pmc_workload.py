"""Workload for the PMC traffic passes (scripts/pmc_traffic.sh): a calibration launch with a known byte count in the
same access pattern as the hot kernels (16 B per lane, coalesced: vsx_normalize on a 1 GiB fp32 buffer reads 1 GiB and
writes 1 GiB), then eager training steps of the bench configuration."""
import argparse
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--mode", default="train", choices=["train", "fwd"], help="fwd: forward-only passes in eval mode (the inference schedule)")
args = ap.parse_args()

from bench import make_batch, nonzero_grn_  # noqa: E402
from viscy_amd._lib import check, lib, ptr, stream  # noqa: E402
from viscy_amd.losses import MixedLoss  # noqa: E402
from viscy_amd.optim import FlatAdamW  # noqa: E402
from viscy_amd.parallel import FlatDataParallel  # noqa: E402
from viscy_amd.step import TrainStep  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

dev = torch.device("cuda", 0)
for mib in (256, 1024):  # calibration: bytes read = bytes written = mib MiB  (kernel: normalize_kernel)
    n = mib * (1 << 20) // 4
    x = torch.rand(n, device=dev)
    y = torch.empty_like(x)
    sub, div = torch.zeros(1, device=dev), torch.ones(1, device=dev)
    for _ in range(2):
        check(lib().vsx_normalize(ptr(x), ptr(y), ptr(sub), ptr(div), 1, n, stream()), "normalize")
    torch.cuda.synchronize()
    del x, y

torch.manual_seed(42)
model = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True).to(dev)
nonzero_grn_(model)
model.compute_dtype = torch.bfloat16
model.grad_mode = "flat"
eng = model.engine()
opt = FlatAdamW(eng, lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=10, warmup_multiplier=1e-3)
ddp = FlatDataParallel(eng, opt)
step = TrainStep(model, MixedLoss(0.5, 0.0, 0.5), opt, ddp, use_graph=False)
x, tgt = make_batch(args.batch, 256, 256, dev, seed=42)
if args.mode == "fwd":
    model.eval()
    with torch.no_grad():
        for _ in range(args.steps):
            y = model(x)
    torch.cuda.synchronize()
    print("fwd mean", float(y.mean()), "passes", args.steps, "batch", args.batch)
else:
    for _ in range(args.steps):
        loss = step(x, tgt)
    torch.cuda.synchronize()
    print("loss", float(loss), "steps", args.steps, "batch", args.batch)
