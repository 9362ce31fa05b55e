"""GPU-only A/B instrument: the bench step (UNeXt2 tiny, Z = 5, 256 x 256, B = 512, bf16, hipGraph replay) timed for the library
`VSX_LIB` selects (viscy_amd/build.py variants) and the flags given as name=value.  One process = one library, so a same-box
comparison is a shell loop `for lib in a b a b; do VSX_LIB=... python tools/ab_step.py; done` (box-to-box spread is 2.5 %,
more than most single changes are worth).

    python tools/ab_step.py [--batch 512] [--steps 10] [--rounds 3] [--ops] [flag=value ...]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from viscy_amd import _lib as L  # noqa: E402
from viscy_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--size", type=int, default=256, help="patch edge (2048 with --batch 8 = the gate shape)")
ap.add_argument("--eager-only", action="store_true", help="N eager steps and nothing else (for a rocprofv3 kernel trace)")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--ops", action="store_true", help="per-(op, shape) event timing of one eager step")
ap.add_argument("--tag", default=os.path.basename(os.environ.get("VSX_LIB", "libvsx.so")))
ap.add_argument("flags", nargs="*")
a = ap.parse_args()
for f in a.flags:
    k, v = f.split("=")
    assert L.lib().vsx_set_flag(k.encode(), int(v)) == 0, f

from viscy_amd.losses import MixedLoss  # noqa: E402
from viscy_amd.optim import FlatAdamW  # noqa: E402
from viscy_amd.parallel import FlatDataParallel  # noqa: E402
from viscy_amd.step import TrainStep  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(42)
model = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4,
               decoder_conv_blocks=2).to(dev)
bench.nonzero_grn_(model)
model.compute_dtype = torch.bfloat16
model.grad_mode = "flat"
eng = model.engine()
opt = FlatAdamW(eng, lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=1000, warmup_multiplier=1e-3)
ddp = FlatDataParallel(eng, opt)
crit = MixedLoss(0.5, 0.0, 0.5)
x, tgt = bench.make_batch(a.batch, a.size, a.size, dev)
eager = TrainStep(model, crit, opt, ddp, use_graph=False)
eager(x, tgt)
if a.eager_only:
    for _ in range(a.steps):
        loss = eager(x, tgt)
    torch.cuda.synchronize()
    print("loss", float(loss))
    sys.exit(0)
if a.ops:
    with bench.OpTimer(ops, by_shape=True) as tm:
        eager(x, tgt)
    rows = sorted(tm.summary().items(), key=lambda kv: -kv[1]["ms"])
    for c, v in rows[:60]:
        print(f"[shape] {c:60s} {v['launches']:3d}x {v['ms'] / v['launches'] * 1e3:9.1f} us {v['ms']:7.2f} ms", flush=True)
graphed = TrainStep(model, crit, opt, ddp, use_graph=True, static_inputs=True)
for _ in range(3):
    loss = graphed(x, tgt)
ms = []
for _ in range(a.rounds):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = graphed(x, tgt)
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) / a.steps * 1e3)
print(json.dumps({"tag": a.tag, "flags": a.flags, "ms_per_step": [round(m, 3) for m in ms], "best": round(min(ms), 3),
                  "loss": float(loss)}), flush=True)
