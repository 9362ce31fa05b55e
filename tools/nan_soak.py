"""GPU-only: hunt the rare non-finite loss of long bf16 training runs (VERDICT r1 "what's weak" 1).

    python tools/nan_soak.py --steps 300 --batch 128 [--poison] [--eager] [--flags nt_stream=3,grn_stream=2,ln_stream=3]

Runs the bench configuration (same model / data / optimiser as bench.py) step by step, checks the loss, the flat gradient
and the flat parameters after EVERY step, keeps a snapshot of (parameters, m, v, t) from before the step, and when a step
goes non-finite
  1. names the parameters whose gradient / value is non-finite (localises the producing kernel),
  2. restores the snapshot and replays that step EAGERLY under ``FiniteGuard`` (first launch whose outputs are
     non-finite) several times — a deterministic reproduction means arithmetic (overflow / 0-division), a
     non-reproducible one means a race / hazard / uninitialised read.
``--poison`` fills every torch.empty allocation with NaN (read-before-write detector; forces --eager).
Prints one JSON summary line at the end.
"""

from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--flags", default="")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--replays", type=int, default=4)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--lr", type=float, default=2e-4)
    args = ap.parse_args()
    if args.flags:
        os.environ["VSX_FLAGS"] = args.flags

    import bench
    from viscy_amd import debug, ops
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.step import TrainStep
    from viscy_amd.unext2 import UNeXt2

    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    model = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True,
                   head_expansion_ratio=4, decoder_conv_blocks=2).to(dev)
    bench.nonzero_grn_(model)
    model.compute_dtype, model.grad_mode = torch.bfloat16, "flat"
    eng = model.engine()
    opt = FlatAdamW(eng, lr=args.lr, schedule="WarmupCosine", warmup_steps=3, t_total=max(args.steps, 4), warmup_multiplier=1e-3)
    crit = MixedLoss(0.5, 0.0, 0.5)
    x, tgt = bench.make_batch(args.batch, args.size, args.size, dev, seed=args.seed)
    use_graph = not (args.eager or args.poison)
    step = TrainStep(model, crit, opt, None, use_graph=use_graph)
    eager = TrainStep(model, crit, opt, None, use_graph=False)
    names = {id(p): n for n, p in model.named_parameters()}

    def bad_params(flat_like):
        out = []
        for p, off in zip(eng.order, eng.offsets):
            sl = flat_like[off:off + p.numel()]
            nb = int((~torch.isfinite(sl)).sum().item())
            if nb:
                out.append((names[id(p)], nb, p.numel()))
        return out

    state = [eng.flat, opt.m, opt.v]
    tsnap = {"t": opt.t}

    def one():
        tsnap["t"] = opt.t
        return step(x, tgt)

    ctx = debug.poison_empty() if args.poison else None
    if ctx:
        ctx.__enter__()
    try:
        res = debug.soak(one, args.steps, state, lambda: {"grad": eng.flat_grad, "param": eng.flat})
    finally:
        if ctx:
            ctx.__exit__(None, None, None)
    summary = {"flags": args.flags, "graph": use_graph, "poison": args.poison, "batch": args.batch, "steps_run": res["steps"],
               "first_bad_step": res["first_bad_step"], "what": res["what"], "loss_first": res["losses"][0],
               "loss_last": res["losses"][-1]}
    if res["first_bad_step"] is not None:
        summary["bad_grads"] = bad_params(eng.flat_grad)[:12]
        summary["n_bad_grad_tensors"] = len(bad_params(eng.flat_grad))
        summary["bad_params"] = bad_params(eng.flat)[:12]
        # replay the failing step eagerly from the snapshot, guarded per launch
        replays = []
        for r in range(args.replays):
            for s, t in zip(res["snapshot"], state):
                t.copy_(s)
            opt.set_step(tsnap["t"])
            g = debug.FiniteGuard(ops, raise_on_first=False)
            with g:
                loss = eager(x, tgt)
            replays.append({"loss": float(loss), "first": g.first, "launches": g.calls})
        summary["replays"] = replays
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
