"""Upper bound of what overlapping launches could buy: TWO independent replicas of the bench step at B = 256 each, captured as
hipGraphs and replayed on two streams at once, against ONE replica at B = 512 (the bench) — the same number of patches.  Every
kernel of the step is sized to fill a CU's registers or LDS, so two launches never share a CU; what concurrency can recover is
the tail of each launch (the last, partly filled round of workgroups) and the drain / ramp between dependent launches.  GPU only."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from viscy_amd.losses import MixedLoss  # noqa: E402
from viscy_amd.optim import FlatAdamW  # noqa: E402
from viscy_amd.step import TrainStep  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

dev = torch.device("cuda:0")
STEPS = int(os.environ.get("STEPS", 10))


def replica(B, seed):
    torch.manual_seed(seed)
    m = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4,
               decoder_conv_blocks=2).to(dev)
    bench.nonzero_grn_(m)
    m.compute_dtype, m.grad_mode = torch.bfloat16, "flat"
    eng = m.engine()
    opt = FlatAdamW(eng, lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=1000, warmup_multiplier=1e-3)
    x, t = bench.make_batch(B, 256, 256, dev, seed=seed)
    return m, opt, x, t


def timed(fn, n=STEPS, rounds=3):
    best = 1e9
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


res = {}
m, opt, x, t = replica(512, 1)
one = TrainStep(m, MixedLoss(0.5, 0.0, 0.5), opt, None, use_graph=True, static_inputs=True)
for _ in range(3):
    one(x, t)
res["one_replica_b512_ms"] = round(timed(lambda: one(x, t)), 3)
del one, m, opt, x, t
import gc  # noqa: E402

gc.collect()
torch.cuda.empty_cache()

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
reps = []
for i, st in enumerate((s1, s2)):
    with torch.cuda.stream(st):
        m, opt, x, t = replica(256, 10 + i)
        step = TrainStep(m, MixedLoss(0.5, 0.0, 0.5), opt, None, use_graph=True, static_inputs=True)
        for _ in range(3):
            step(x, t)
    torch.cuda.synchronize()
    reps.append((step, x, t, st))


def both_concurrent():
    for step, x, t, st in reps:
        with torch.cuda.stream(st):
            step(x, t)


def both_serial():
    for step, x, t, st in reps:
        with torch.cuda.stream(s1):
            step(x, t)


res["two_replicas_b256_same_stream_ms"] = round(timed(both_serial), 3)
res["two_replicas_b256_two_streams_ms"] = round(timed(both_concurrent), 3)
print(json.dumps(res))
