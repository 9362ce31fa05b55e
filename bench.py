#!/usr/bin/env python
"""bench.py — training throughput of the UNeXt2 virtual-staining hot path on MI355X.

Workload (BASELINE.json configs[1]): UNeXt2 2.5D (convnextv2_tiny, Z=5, 256x256, 1→2 ch, head_pool,
2 decoder blocks/stage) bf16 training on synthetic patches; one "step" = forward + MixedLoss(0.5, 0, 0.5)
+ backward + AdamW on one batch, all in hand-written HIP kernels.  Inputs are resident in HBM before
the timed region.  N > 1: one process per GPU (torch.distributed.run), batch sharded (weak scaling),
RCCL all-reduce of the flat gradient buffer in three buckets: the step is captured as hipGraph segments that end where a
bucket completes, and each bucket's all-reduce is issued between two replays, under the next segment (viscy_amd/step.py).

Prints ONE JSON line (rank 0) with the contract fields + "roofline" (dominant kernel class, timed
live with HIP events on the launch stream inside the timed region) + "cpu_baseline" (the oracle —
pure-torch fp32 restatement of the reference — timed on the host cores over a bounded sample).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_PATCH = 22.56  # SURVEY §8d / BASELINE.md §2 (256x256, Z=5, tiny)
FWD_MB_PER_PATCH = 75.5  # forward "two-pass floor" 
ALGO_MB_PER_PATCH = 226.5    # fwd+bwd two-pass-GRN floor, bf16 activations
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0


def source_hash() -> str:
    """identity of the kernel sources this process runs (sha256 over viscy_amd/csrc/* and include/vsx.h): profiles and
    PMC traffic files carry it, and a file measured on other sources is refused"""
    import hashlib

    h = hashlib.sha256()
    files = sorted(os.path.join("viscy_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "viscy_amd", "csrc"))) + [os.path.join("include", "vsx.h")]
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def build_info() -> dict:
    """git SHA recorded by viscy_amd.build at build time (the GPU box has no .git), source hash, and every kernel flag"""
    from viscy_amd import _lib

    info = {"git_sha": None, "git_dirty": None}
    try:
        info.update(json.load(open(os.path.join(ROOT, "viscy_amd", "_build", "build_info.json"))))
    except (OSError, ValueError):
        pass
    info["source_hash"] = source_hash()
    l = _lib.lib()
    info["flags"] = {n: int(l.vsx_get_flag(n.encode())) for n in FLAG_NAMES}
    return info


FLAG_NAMES = ["tn_tr", "nt_wide", "nt_fast", "tn_wide", "nt_tall", "nt2", "nt_stream", "grn_stream", "ggb_contig", "tn_want", "tn_contig",
              "ln_stream", "ggb_blocks", "tn_rect", "dw_rows2", "dw_wg16", "dw_mfma", "ln_fblk", "ln_bblk", "mlp_fused", "loss_fused"]


def make_batch(B, H, W, device, seed=42):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, 1, 5, H, W), generator=g)
    smooth = torch.nn.functional.avg_pool3d(x, (1, 5, 5), stride=1, padding=(0, 2, 2))
    tgt = 0.5 * smooth.repeat(1, 2, 1, 1, 1) + 0.1 * torch.randn((B, 2, 5, H, W), generator=g)
    return x.to(device), tgt.contiguous().to(device)


def nonzero_grn_(model, seed=1):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".grn." in n:
                p.copy_((torch.randn(p.shape, generator=g) * 0.1).to(p.device))


# ------------------------------------------------------------------ per-op event timing
class OpTimer:
    """Wraps viscy_amd.ops launch wrappers with torch.cuda events (recorded on the current stream,
    which is the stream every kernel is launched on)."""

    GEMM_EPI = {0: "none", 1: "bias", 2: "bias_gelu_sq", 3: "bias_res", 4: "dz", 5: "bias_stats"}

    def __init__(self, ops, only: str | None = None, by_shape: bool = False):
        self.ops, self.only, self.by_shape = ops, only, by_shape
        self.records = {}  # name -> list of (start, end, flops, bytes)
        self._orig = {}

    @staticmethod
    def _bytes(args, out):
        n = 0
        seen = set()
        stack = list(args) + (list(out) if isinstance(out, (tuple, list)) else [out])
        for t in stack:
            if torch.is_tensor(t) and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                n += t.numel() * t.element_size()
        return n

    def _wrap(self, name, fn):
        def wrapped(*a, **k):
            cls, flops, nbytes, strict = name, 0.0, None, None
            if name == "gemm":
                kind, M, N, K = a[0], a[4], a[5], a[6]
                nz = k.get("nz", 1)
                es = 2 if k.get("dtype") == torch.bfloat16 else 4
                cls = f"gemm_{kind}" if not self.by_shape else f"gemm_{kind} M{M} N{N} K{K} z{nz} e{k.get('epi', 0)} p{k.get('pro', 0)} a{k.get('a_mode', 0)}"
                flops = 2.0 * M * N * K * nz
                nbytes = (M * K + M * N) * es * nz + N * K * (es if kind == "nt" else 4)
                strict = nbytes
                if kind == "nt":
                    # operands of the fused epilogues are part of the launch's algorithmic traffic: the second output of
                    # fc1 (g = gelu(h)), the activation the dZ epilogue reads for the GRN statistics, the residual of fc2
                    extra = sum(k.get(name) is not None for name in ("C2", "aux", "res")) - (a[3] is None)
                    nbytes += extra * M * N * es * nz
            elif name in ("mlp_stats", "mlp_out"):  # fused GRN-MLP: M = a[-3], C = a[-2]
                Mm, Cc = a[-3], a[-2]
                flops = 2.0 * Mm * 4 * Cc * Cc * (2 if name == "mlp_out" else 1)
                nbytes = Mm * Cc * 2 * (3 if name == "mlp_out" else 1) + 16 * Cc * Cc
            if self.only is not None and cls != self.only:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            if nbytes is None:
                nbytes = self._bytes(list(a) + list(k.values()), out)
            self.records.setdefault(cls, []).append((e0, e1, flops, nbytes, strict if strict is not None else nbytes))
            return out

        return wrapped

    def __enter__(self):
        for name in dir(self.ops):
            fn = getattr(self.ops, name)
            if callable(fn) and not name.startswith("_") and getattr(fn, "__module__", "") == self.ops.__name__ and name not in ("gemm_z",):
                self._orig[name] = fn
                setattr(self.ops, name, self._wrap(name, fn))
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(self.ops, name, fn)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for cls, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            out[cls] = {"launches": len(recs), "ms": ms, "flops": sum(r[2] for r in recs), "bytes": sum(r[3] for r in recs),
                        "strict_bytes": sum(r[4] for r in recs)}
        return out


def gate_shape_record(dev, steps: int = 5, B: int = 8, size: int = 2048) -> dict:
    """training step at the north star's gate shape (B = 8 per GPU, Z = 5, 2048 x 2048, bf16): ms / step and both roofline
    fractions priced with SURVEY §8(d)'s per-sample figures (x64 the 256^2 patch: 4.83 GB two-pass floor, 1.444 TFLOP fwd)"""
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.step import TrainStep
    from viscy_amd.unext2 import UNeXt2

    torch.manual_seed(42)
    model = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True,
                   head_expansion_ratio=4, decoder_conv_blocks=2).to(dev)
    nonzero_grn_(model)
    model.compute_dtype, model.grad_mode = torch.bfloat16, "flat"
    opt = FlatAdamW(model.engine(), lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=steps + 4, warmup_multiplier=1e-3)
    x, tgt = make_batch(B, size, size, dev, seed=7)
    step = TrainStep(model, MixedLoss(0.5, 0.0, 0.5), opt, None, use_graph=True, static_inputs=True)
    l0 = step(x, tgt).clone()  # (the step returns its static loss tensor: keep the first value)
    step(x, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x, tgt)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / steps
    scale = (size / 256.0) ** 2
    sps = B / dt_s
    return {"workload": f"UNeXt2 tiny Z=5 {size}x{size} 1->2ch, B={B}, fwd+MixedLoss+bwd+AdamW (hipGraph)", "steps": steps,
            "ms_per_step": round(dt_s * 1e3, 3), "stacks_per_s": round(sps, 2), "patch_equivalents_per_s": round(sps * scale, 1),
            "hbm_frac_of_algorithmic_floor": round(sps * ALGO_MB_PER_PATCH * scale * 1e6 / (HBM_PEAK_GBS * 1e9), 4),
            "mfma_frac": round(sps * 3 * FWD_GFLOP_PER_PATCH * scale * 1e9 / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
            "first_loss": round(float(l0), 5), "final_loss": round(float(loss), 5), "loss_finite": bool(torch.isfinite(loss).item()),
            "peak_hbm_gb": round(torch.cuda.max_memory_reserved() / 1e9, 1)}


def _cpu_baseline_child():
    """Runs in a child process; prints one JSON line per completed measurement (the parent keeps the last)."""
    from oracle import loss_ref, unext2_ref

    # torch's CPU kernels stop scaling (and then regress) well below the 256 hardware threads of the GPU host:
    # 32 threads measured fastest there (tools/cpu_probe.py); `cores` in the JSON is the thread count actually used
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True)
    model = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=0)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
    B = 2
    x, tgt = make_batch(B, 256, 256, "cpu")
    times = []
    for i in range(12):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = loss_ref.mixed_loss(model(x), tgt, 0.5, 0.0, 0.5)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if i >= 1 or dt > 8.0:  # first iteration is warm-up unless the host is so slow that one step is all we get
            times.append(dt)
            ts = sorted(times)
            med = ts[len(ts) // 2]
            print(json.dumps({"value": round(B / med, 3), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
                              "sample": f"{len(times)} timed training step(s) (fwd+MixedLoss+bwd+AdamW) of the fp32 oracle "
                                        f"(pure-torch restatement of the reference) at B={B}, Z=5, 256x256, median"}), flush=True)


def cpu_baseline(budget_s: float = 75.0):
    """The oracle timed on the host cores over a bounded sample (child process, hard time limit)."""
    import subprocess

    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], capture_output=True, text=True,
                           timeout=budget_s)
        out = p.stdout
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if not lines:
        return {"value": None, "unit": "patches/s", "cores": os.cpu_count(), "kind": "port",
                "sample": f"no oracle training step finished within {budget_s:.0f} s on this host"}
    return json.loads(lines[-1])


def main():
    if "--cpu-baseline-child" in sys.argv:
        _cpu_baseline_child()
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("VSX_BENCH_BATCH", 512)), help="patches per GPU per step")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gate", action="store_true", help="skip the (B=8, 2048^2) gate-shape sub-record")
    ap.add_argument("--gate-steps", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured step")
    ap.add_argument("--profile-ops", action="store_true", help="print the per-op-class event-timing table to stderr")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel collective path even with one rank (RCCL smoke test on a single GPU)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: everything any library writes to file descriptor 1 during the run
    # (RCCL prints a version banner through C stdio at communicator init and flushes it at exit, i.e. AFTER a Python print)
    # is routed to stderr, and the result is written to the saved descriptor at the very end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_dp:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from viscy_amd import ops
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.parallel import FlatDataParallel
    from viscy_amd.unext2 import UNeXt2

    torch.manual_seed(42)
    model = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True,
                   head_expansion_ratio=4, decoder_conv_blocks=2).to(dev)
    nonzero_grn_(model)
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.grad_mode = "flat"
    eng = model.engine()
    total_steps = args.steps + args.warmup
    opt = FlatAdamW(eng, lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=max(total_steps, 4), warmup_multiplier=1e-3)
    ddp = FlatDataParallel(eng, opt, force=args.force_dp)
    crit = MixedLoss(0.5, 0.0, 0.5)
    B = args.batch
    x, tgt = make_batch(B, args.size, args.size, dev, seed=42 + rank)

    from viscy_amd.step import TrainStep

    eager = TrainStep(model, crit, opt, ddp, use_graph=False)
    # the batch is resident in HBM before the timed region (the contract): the captured step reads it in place instead of
    # copying it into buffers of its own first (TrainStep.static_inputs)
    graphed = TrainStep(model, crit, opt, ddp, use_graph=not args.no_graph, static_inputs=True)

    def step():
        return graphed(x, tgt)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- warmup (the last warmup step is instrumented per op class to find the dominant kernel)
    if args.profile_ops:
        with OpTimer(ops, by_shape=True) as tm:
            eager(x, tgt)
        if rank == 0:
            for c, v in sorted(tm.summary().items(), key=lambda kv: -kv[1]["ms"])[:40]:
                print(f"[shape] {c:58s} {v['launches']:3d}x {v['ms'] / v['launches'] * 1e3:9.1f} us  "
                      f"{v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} GB/s {v['flops'] / max(v['ms'], 1e-9) / 1e9:8.1f} TFLOP/s", file=sys.stderr)
    if not args.profile_ops:
        eager(x, tgt)  # first launches load code objects / size workspaces: keep that out of the instrumented step
    with OpTimer(ops) as tm:  # one eager, instrumented step: per-op-class times → dominant kernel class
        l0 = eager(x, tgt)
    table = tm.summary()
    for _ in range(max(args.warmup, 1)):
        step()
    dominant = max(table, key=lambda c: table[c]["ms"]) if table else None
    if args.profile_ops and rank == 0:
        tot = sum(v["ms"] for v in table.values())
        for c, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            print(f"[ops] {c:28s} {v['launches']:5d} launches {v['ms']:9.3f} ms {100 * v['ms'] / tot:5.1f}%  "
                  f"{v['bytes'] / max(v['ms'], 1e-9) / 1e6:9.1f} GB/s {v['flops'] / max(v['ms'], 1e-9) / 1e9:9.1f} TFLOP/s", file=sys.stderr)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # dominant kernel class: average launch duration measured live with HIP events on the launch stream, over
    # eager replays of the same step right behind the timed region (events cannot be recorded inside a graph replay)
    with OpTimer(ops, only=dominant) as tm:
        for _ in range(3):
            eager(x, tgt)
    # the metric's second half, "fwd HBM GB/s": forward-only passes (inference schedule, one hipGraph replay each) priced at
    # SURVEY §8(d)'s algorithmic two-pass floor of 75.5 MB per patch
    from viscy_amd.step import InferStep

    model.eval()
    infer = InferStep(model)
    nf = max(3, min(args.steps, 10))
    infer(x)
    barrier()
    tf0 = time.perf_counter()
    for _ in range(nf):
        infer(x)
    barrier()
    fwd_s = (time.perf_counter() - tf0) / nf
    model.train()
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()
    dom = tm.summary().get(dominant) if dominant else None

    binfo = build_info()
    gate, peak_main = None, None
    if rank == 0 and world == 1 and args.size == 256 and args.dtype == "bf16" and not args.no_gate:
        # the north star's roofline gate shape, driver-timed in the same run: (B = 8, Z = 5, 2048 x 2048) training step.
        # The headline model / graph / batch are released first (both workloads need ~166 GB of the 288 GB).
        loss_keep = loss.detach().clone()
        peak_main = torch.cuda.max_memory_reserved()
        del graphed, eager, infer, tm, loss, step
        model._engine = None
        opt = ddp = eng = None
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        gate = gate_shape_record(dev, steps=args.gate_steps)
        loss = loss_keep
    if rank == 0:
        patches = world * B * args.steps
        value = patches / elapsed
        scale = (args.size / 256.0) ** 2
        roof = None
        if dom:
            ms = dom["ms"] / dom["launches"]
            tf = dom["flops"] / dom["launches"] / (ms * 1e-3) / 1e12
            gbs = dom["bytes"] / dom["launches"] / (ms * 1e-3) / 1e9
            if dominant.startswith("gemm") and tf / MFMA_BF16_PEAK_TFLOPS >= gbs / HBM_PEAK_GBS:
                roof = {"kernel": dominant, "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                        "avg_launch_ms": round(ms, 5), "launches": dom["launches"],
                        "hbm_side_GBs": round(gbs, 1)}
            else:
                roof = {"kernel": dominant, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(ms, 5),
                        "launches": dom["launches"]}
        if roof is not None and dom:
            roof["strict_frac"] = round(dom["strict_bytes"] / dom["launches"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)  # operands + ONE output only
            # HBM traffic of the dominant kernel class: PMC counters cannot be read from inside this process, so the
            # figure comes from the committed rocprofv3 --pmc passes of the SAME configuration (scripts/pmc_traffic.sh ->
            # profiles/r03_pmc_traffic_b<batch>.json: FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on a launch with
            # a known byte count as MI355X_MICROARCH.md prescribes).  The file records the kernel-source hash and the flag
            # set it was measured with; a file from other sources / flags is refused (traffic stays null).
            tf_path = os.path.join(ROOT, "profiles", f"r03_pmc_traffic_b{B}.json")
            if args.size == 256 and args.dtype == "bf16" and os.path.exists(tf_path):
                try:
                    tfj = json.load(open(tf_path))
                    same = tfj.get("source_hash") == binfo["source_hash"] and tfj.get("flags") == binfo["flags"]
                    cls = tfj.get("classes", {}).get(dominant)
                    if cls and same:
                        roof["traffic"] = round(cls["traffic_bytes_per_launch"])
                        roof["algorithmic_bytes_per_launch"] = round(dom["bytes"] / dom["launches"])
                        roof["traffic_source"] = os.path.relpath(tf_path, ROOT)
                    elif cls:
                        roof["traffic_refused"] = "profiles file was measured on other kernel sources / flags"
                except (OSError, ValueError, KeyError):
                    pass
        res = {
            "metric": "training patches/sec (Z=5, 256x256, 1->2ch UNeXt2)",
            "value": round(value, 2),
            "unit": "patches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"UNeXt2 2.5D convnextv2_tiny Z=5 {args.size}x{args.size} 1->2ch, fwd+MixedLoss(0.5,0,0.5)+bwd+AdamW",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "whole_path": {
                "hbm_frac_of_algorithmic_floor": round(value * ALGO_MB_PER_PATCH * scale * 1e6 / (world * HBM_PEAK_GBS * 1e9), 4),
                "mfma_frac": round(value * 3 * FWD_GFLOP_PER_PATCH * scale * 1e9 / (world * MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
                "final_loss": round(float(loss.item()), 5), "first_loss": round(float(l0.item()), 5),
                "loss_finite": bool(torch.isfinite(loss).item()),  # a non-finite loss invalidates the line
                "peak_hbm_gb": round((peak_main if peak_main is not None else torch.cuda.max_memory_reserved()) / 1e9, 1),
            },
            "roofline": roof,
            "gate_shape": gate,
            "build": binfo,
            "fwd": {"ms_per_pass": round(fwd_s * 1e3, 3), "patches_per_s_per_gpu": round(B / fwd_s, 1),
                    "algorithmic_hbm_GBps": round(B / fwd_s * FWD_MB_PER_PATCH * scale / 1e3, 1),
                    "frac_hbm_peak": round(B / fwd_s * FWD_MB_PER_PATCH * scale * 1e6 / (HBM_PEAK_GBS * 1e9), 4),
                    "tflops": round(B / fwd_s * FWD_GFLOP_PER_PATCH * scale / 1e3, 1)},
        }
        if not res["whole_path"]["loss_finite"]:
            print("bench.py: WARNING: the training loss is not finite after the timed steps — this measurement is INVALID",
                  file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only (the other ranks would idle)
            res["cpu_baseline"] = cpu_baseline()
    if world > 1 or args.force_dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    os.close(result_fd)


if __name__ == "__main__":
    main()
