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
PROFILE_ROUND = "r06"        # prefix of this round's files under profiles/
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


FLAG_NAMES = ["tn_tr", "nt_wide", "nt_fast", "tn_wide", "nt2", "nt_stream", "grn_stream", "ggb_contig", "tn_want", "tn_contig", "tn_stream",
              "ln_stream", "ggb_blocks", "tn_rect", "dw_mfma", "ln_fblk", "ln_bblk", "ln_ablk", "mlp_fused", "loss_fused", "mlp_sf32", "head_rows", "det_reduce"]


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


# ------------------------------------------------------------------ per-family event timing
# Kernel FAMILIES (one per kernel template / kernel group), named the same way from both sides: the launch wrappers of
# viscy_amd.ops that bench.py times live with HIP events (OP_FAMILY; the GEMM entry points report the template they dispatched
# through vsx_last_kernel()), and the kernel names of a rocprofv3 trace (KERNEL_FAMILY, used by tools/roofline_table.py and
# tools/pmc_traffic.py) — so that the live table, the kernel-stats CSV and the PMC traffic file can be laid side by side.
OP_FAMILY = {
    "mlp_stats": "mlp_fused", "mlp_out": "mlp_fused", "mlp_fc1": "mlp_fused", "mlp_fc1_ln": "mlp_fused", "mlp_bwd_stats": "mlp_fused",
    "mlp_bwd_dh": "mlp_fused", "mlp_bwd_dh_re": "mlp_fused", "mlp_bwd_dh_ln": "mlp_fused",
    "dwconv7_fwd": "dwconv7", "dwconv7_bwd_data": "dwconv7", "dwconv7_bwd_weight": "dwconv7", "dwconv7_bwd": "dwconv7",
    "head_shuffle_fwd": "head", "head_shuffle_bwd": "head", "head_out_fwd": "head", "head_out_bwd1": "head", "head_out_bwd1_wgrad": "head",
    "head_out_bwd2": "head", "head_conv_fwd": "head", "head_conv_wgrad": "head", "head_conv_dgrad": "head", "head_conv_dgrad_prep": "head",
    "prep_head_dgrad": "head",
    "ln_fwd": "layernorm", "ln_bwd": "layernorm",
    "grn_scale": "grn_small", "grn_bwd_stats": "grn_small", "grn_q_reduce": "grn_small", "grn_gelu_bwd": "grn_gelu_bwd",
    "scale_weight_samples": "grn_small",
    "pixel_shuffle_cat_fwd": "ps_cat", "pixel_shuffle_cat_bwd": "ps_cat", "stem_im2col": "stem_im2col",
    "flush": "weight_tasks", "adamw": "adamw",
}
# wrappers that only delegate to other wrappers (timing them would count their launches twice) or launch nothing
OP_SKIP = {"gemm_z", "dgrad_ln_bwd", "zeros", "batch", "batch_open", "batch_close", "mlp_supported", "head_conv_supported"}
KERNEL_FAMILY = [
    (r"mlp_fused_kernel", "mlp_fused"), (r"gemm_nt2_(lnbwd_)?kernel", "gemm_nt2"), (r"gemm_nt_fast_kernel", "gemm_nt_fast"),
    (r"gemm_nt_kernel", "gemm_nt_generic"), (r"gemm_tn_fast_kernel", "gemm_tn_fast"), (r"gemm_tn_kernel|tn_zero_kernel", "gemm_tn_generic"),
    (r"dwconv7|dw_reduce_rows", "dwconv7"), (r"head_", "head"), (r"ssim_|loss_", "loss"), (r"ln_(fwd|bwd)_kernel", "layernorm"),
    (r"grn_gelu_bwd", "grn_gelu_bwd"), (r"grn_|scale_weight_samples", "grn_small"), (r"ps_cat", "ps_cat"), (r"stem_im2col", "stem_im2col"),
    (r"weight_tasks", "weight_tasks"), (r"adamw", "adamw"),
]


def kernel_family(kernel_name: str) -> str:
    import re

    for pat, fam in KERNEL_FAMILY:
        if re.search(pat, kernel_name):
            return fam
    return "other"


class OpTimer:
    """Wraps viscy_amd.ops launch wrappers with torch.cuda events (recorded on the current stream,
    which is the stream every kernel is launched on) and groups the launches by kernel family."""

    def __init__(self, ops, only: str | None = None, by_shape: bool = False):
        self.ops, self.only, self.by_shape = ops, only, by_shape
        self.records = {}    # family -> list of (start, end, flops, bytes, strict bytes, floor bytes)
        self.templates = {}  # kernel template (family + the template arguments / GEMM shape that select the code) -> same tuples
        self._orig = {}

    @staticmethod
    def _tensors(args, out):
        seen, ts = set(), []
        stack = list(args) + (list(out) if isinstance(out, (tuple, list)) else [out])
        for t in stack:
            if torch.is_tensor(t) and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                ts.append(t)
        return ts

    def _wrap(self, name, fn):
        import inspect

        sig = inspect.signature(fn)

        def wrapped(*a, **k):
            cls, flops, nbytes, strict, floor = OP_FAMILY.get(name, name), 0.0, None, None, None
            is_gemm = name == "gemm"
            if is_gemm:
                kind, M, N, K = a[0], a[4], a[5], a[6]
                nz = k.get("nz", 1)
                es = 2 if k.get("dtype") == torch.bfloat16 else 4
                flops = 2.0 * M * N * K * nz
                nbytes = (M * K + M * N) * es * nz + N * K * (es if kind == "nt" else 4)
                strict = nbytes
                # SURVEY §8(d)'s floor counts a ConvNeXt block as "read x, write y (+ the GRN second read)": the 4C-wide hidden
                # tensors (g, dh: the [M, max(N, K)] operand of an fc1- / fc2-shaped product, N = 4K or K = 4N) and the per-sample
                # fp32 products Q_b are DESIGN bytes of this implementation, not floor bytes
                wide = max(N, K) == 4 * min(N, K)
                floor = (M * min(N, K) if wide else M * K + M * N) * es * nz + (0 if k.get("b_bstride") and kind == "tn" else N * K * (es if kind == "nt" else 4))
                if kind == "nt":
                    # operands of the fused epilogues are part of the launch's algorithmic traffic: the second output of
                    # fc1 (g = gelu(h)), the activation the dZ epilogue reads for the GRN statistics, the residual of fc2
                    extra = sum(k.get(nm) is not None for nm in ("C2", "aux", "res")) - (a[3] is None)
                    nbytes += extra * M * N * es * nz
                    if not (wide and N > K):  # a C-wide second operand of the epilogue (residual, LayerNorm-backward rows)
                        floor += extra * M * N * es * nz
            elif cls == "mlp_fused":  # fused GRN-MLP passes: one (two for the output pass) M x 4C x C contraction(s)
                ba = sig.bind(*a, **k).arguments
                Mm, Cc = ba["M"], ba["C"]
                flops = 2.0 * Mm * 4 * Cc * Cc * (2 if name in ("mlp_out", "mlp_bwd_dh_re", "mlp_bwd_dh_ln") else 1)
            if self.only is not None and not is_gemm and cls != self.only:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            tmpl = name
            if is_gemm:
                cls = lib_last_kernel() or f"gemm_{a[0]}"
                tmpl = cls + f" M{M} N{N} K{K} z{nz} e{k.get('epi', 0)} p{k.get('pro', 0)} a{k.get('a_mode', 0)}"
                if self.by_shape:
                    cls = tmpl
                if self.only is not None and cls != self.only:
                    return out
            elif cls == "mlp_fused":
                tmpl = f"mlp_fused_kernel<C={Cc}, MODE={mlp_mode(name, ba)}>"
            elif "C" in sig.parameters:
                try:
                    tmpl = f"{name} C={sig.bind(*a, **k).arguments['C']}"
                except TypeError:
                    pass
            if nbytes is None:
                ts = self._tensors(list(a) + list(k.values()), out)
                nbytes = sum(t.numel() * t.element_size() for t in ts)
                if cls == "mlp_fused":
                    # SURVEY §8(d)'s floor counts a block as "read x, write y (+ the GRN second read)": the 4C-wide h / g / dh
                    # this kernel family moves are design bytes, not floor bytes
                    floor = sum(t.numel() * t.element_size() for t in ts if t.numel() != Mm * 4 * Cc)
            rec = (e0, e1, flops, nbytes, strict if strict is not None else nbytes, floor if floor is not None else nbytes)
            self.records.setdefault(cls, []).append(rec)
            self.templates.setdefault(tmpl, []).append(rec + (cls,))
            return out

        return wrapped

    def __enter__(self):
        for name in dir(self.ops):
            fn = getattr(self.ops, name)
            if callable(fn) and not name.startswith("_") and getattr(fn, "__module__", "") == self.ops.__name__ and name not in OP_SKIP:
                self._orig[name] = fn
                setattr(self.ops, name, self._wrap(name, fn))
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(self.ops, name, fn)

    def summary(self, templates: bool = False):
        torch.cuda.synchronize()
        out = {}
        for cls, recs in (self.templates if templates else self.records).items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            out[cls] = {"launches": len(recs), "ms": ms, "flops": sum(r[2] for r in recs), "bytes": sum(r[3] for r in recs),
                        "strict_bytes": sum(r[4] for r in recs), "floor_bytes": sum(r[5] for r in recs)}
            if templates:
                out[cls]["family"] = recs[0][6]
        return out


def mlp_mode(op_name: str, bound_args: dict) -> int:
    """MODE template argument of csrc/mlp.hip that a viscy_amd.ops wrapper of the fused GRN-MLP family launches"""
    if op_name in ("mlp_fc1", "mlp_fc1_ln"):
        return 2 if bound_args.get("store_h", True) else 6
    return {"mlp_stats": 0, "mlp_out": 1, "mlp_bwd_stats": 3, "mlp_bwd_dh": 4, "mlp_bwd_dh_re": 5, "mlp_bwd_dh_ln": 7}[op_name]


def lib_last_kernel() -> str:
    from viscy_amd import _lib

    return _lib.lib().vsx_last_kernel().decode()


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
    opt = FlatAdamW(model.engine(), lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=steps + 7, warmup_multiplier=1e-3)
    x, tgt = make_batch(B, size, size, dev, seed=7)
    step = TrainStep(model, MixedLoss(0.5, 0.0, 0.5), opt, None, use_graph=True, static_inputs=True)
    l0 = step(x, tgt).clone()  # (the step returns its static loss tensor: keep the first value)
    for _ in range(4):  # untimed replays: the first ones behind a capture run 3 - 4 % slow (tools/ab_step.py on the same box: 91.9 ms
        step(x, tgt)    # where a sub-run timed from its second replay reported 95.6)
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


def _cpu_baseline_child(fixture_path: str | None = None, only_threads: int | None = None):
    """Runs in a child process; prints one JSON line per completed measurement (the parent keeps the last).  Phase 1 writes the
    fp32 parity fixture; the timing phase starts only when the parent has finished its GPU sub-records (`<fixture>.go` appears):
    a multi-threaded oracle step beside the gate-shape / fp32 sub-runs perturbed both sides."""
    from oracle import loss_ref, unext2_ref

    ncpu = os.cpu_count() or 1
    if only_threads:  # the all-cores leg: a FRESH process that starts with that many threads (re-sizing the thread pool of a
        fixture_path = None  # process that already ran at 32 threads did not finish one step in 87 s on the 256-thread host)
    # torch's CPU kernels stop scaling (and then regress) well below the 256 hardware threads of the GPU host: 32 threads
    # measured fastest there (tools/cpu_probe.py).  BOTH figures are reported: `value` / `cores` = the 32-thread
    # run, `all_cores` = the same step under set_num_threads(os.cpu_count()), which is what BASELINE.md section 3 prescribes.
    torch.set_num_threads(only_threads or min(ncpu, 32))
    kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True)
    model = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=0)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
    B = 2
    x, tgt = make_batch(B, 256, 256, "cpu")
    if fixture_path:
        # the oracle as CHECKER of the fp32 parity engine (`fp32_parity.forward_rel_err` of the result line): its weights, one
        # input batch and its fp32 forward go to a scratch file that the GPU process compares its own forward with
        with torch.no_grad():
            y_ref = model(x)
        torch.save({"state_dict": model.state_dict(), "x": x, "y": y_ref}, fixture_path)
        print(json.dumps({"fixture": fixture_path}), flush=True)
        t_end = time.perf_counter() + 900.0
        while not os.path.exists(fixture_path + ".go") and time.perf_counter() < t_end:
            time.sleep(0.2)

    def run(threads: int, n_steps: int, budget_s: float, extra: dict) -> dict | None:
        torch.set_num_threads(threads)
        times, rec, t_start = [], None, time.perf_counter()
        for i in range(n_steps + 1):
            t0 = time.perf_counter()
            opt.zero_grad()
            loss = loss_ref.mixed_loss(model(x), tgt, 0.5, 0.0, 0.5)
            loss.backward()
            opt.step()
            dt = time.perf_counter() - t0
            if i >= 1 or dt > 8.0:  # first iteration is warm-up unless the host is so slow that one step is all we get
                times.append(dt)
                med = sorted(times)[len(times) // 2]
                rec = {"value": round(B / med, 3), "unit": "patches/s", "cores": threads, "threads_used": threads,
                       "host_cores": ncpu, "kind": "port",
                       "sample": f"{len(times)} timed training step(s) (fwd+MixedLoss+bwd+AdamW) of the fp32 oracle "
                                 f"(pure-torch restatement of the reference) at B={B}, Z=5, 256x256, median; timed after the GPU "
                                 "sub-records, host otherwise idle"}
                print(json.dumps({**rec, **extra}), flush=True)
            if time.perf_counter() - t_start > budget_s:
                break
        return rec

    if only_threads:
        run(only_threads, 4, 30.0, {"_partial": True})
        return
    first = run(min(ncpu, 32), 10, 25.0, {})
    if first is not None and ncpu > 32:
        import subprocess

        print(json.dumps({"_all_cores_started": time.time(), "threads": ncpu}), flush=True)
        env = dict(os.environ, OMP_NUM_THREADS=str(ncpu), MKL_NUM_THREADS=str(ncpu))
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "-", "--threads", str(ncpu)],
                                 capture_output=True, text=True, timeout=60.0, env=env).stdout
        except subprocess.TimeoutExpired as e:
            out = (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        vals = [json.loads(l) for l in out.splitlines() if l.startswith("{") and '"value"' in l]
        if vals:
            first["all_cores"] = {"value": vals[-1]["value"], "threads_used": ncpu,
                                  "note": "same step in a fresh process under torch.set_num_threads(os.cpu_count()), BASELINE.md section 3: "
                                          + vals[-1]["sample"].split(" of the fp32")[0]}
            print(json.dumps(first), flush=True)


class CpuBaseline:
    """The oracle timed on the host cores over a bounded sample: a child process with a hard time limit.  It is started early
    for the fp32 parity fixture (`fixture()`), then WAITS; `result()` releases its timing phase — after the gate-shape and fp32
    sub-records are done, so that neither side competes for host cores — and waits for the timing lines."""

    def __init__(self, budget_s: float = 100.0):
        import subprocess
        import tempfile

        self.budget_s = budget_s
        self.fixture_path = os.path.join(tempfile.mkdtemp(prefix="vsx_bench_"), "fp32_parity.pt")
        self.out_path = self.fixture_path + ".log"
        self._log = open(self.out_path, "w")
        self.p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", self.fixture_path],
                                  stdout=self._log, stderr=subprocess.DEVNULL, text=True)

    def _lines(self):
        try:
            return [l for l in open(self.out_path).read().splitlines() if l.startswith("{")]
        except OSError:
            return []

    def fixture(self, wait_s: float = 60.0):
        """path of the oracle's (weights, input, fp32 forward) file, or None if the child did not get that far in time"""
        t_end = time.perf_counter() + wait_s
        while time.perf_counter() < t_end:
            if any('"fixture"' in l for l in self._lines()):
                return self.fixture_path
            if self.p.poll() is not None:
                break
            time.sleep(0.25)
        return self.fixture_path if any('"fixture"' in l for l in self._lines()) else None

    def result(self) -> dict:
        self.fixture(60.0)
        open(self.fixture_path + ".go", "w").close()
        try:
            self.p.wait(timeout=self.budget_s)
        except Exception:  # noqa: BLE001 — subprocess.TimeoutExpired: the bounded sample ends here
            self.p.kill()
            self.p.wait()
        self._t_end = time.time()
        self._log.close()
        lines = [json.loads(l) for l in self._lines() if '"value"' in l]
        for f in (self.fixture_path, self.fixture_path + ".go"):
            try:
                os.remove(f)
            except OSError:
                pass
        full = [l for l in lines if not l.get("_partial")]
        if not full:
            return {"value": None, "unit": "patches/s", "cores": min(os.cpu_count() or 1, 32), "threads_used": min(os.cpu_count() or 1, 32),
                    "host_cores": os.cpu_count(), "kind": "port",
                    "sample": f"no oracle training step finished within {self.budget_s:.0f} s on this host"}
        res = full[-1]
        part = [l for l in lines if l.get("_partial")]
        if "all_cores" not in res and part:  # the all-cores run was cut by the time limit: keep what it had
            res["all_cores"] = {"value": part[-1]["value"], "threads_used": part[-1]["threads_used"],
                                "note": "same step under torch.set_num_threads(os.cpu_count()), cut by the time limit"}
        started = [json.loads(l) for l in self._lines() if "_all_cores_started" in l]
        if "all_cores" not in res and started:  # not one step finished: an upper bound is still a measurement
            waited = self._t_end - started[-1]["_all_cores_started"]
            res["all_cores"] = {"value": None, "threads_used": started[-1]["threads"], "upper_bound": round(2.0 / max(waited, 1e-3), 3),
                                "note": f"no training step finished within {waited:.0f} s under torch.set_num_threads(os.cpu_count()) "
                                        f"(BASELINE.md section 3): below {2.0 / max(waited, 1e-3):.3f} patches/s, i.e. slower than the "
                                        "32-thread figure above — torch's CPU kernels regress beyond ~32 threads on this host"}
        return res


def fp32_parity_record(dev, fixture_path: str | None, B: int = 128, steps: int = 3) -> dict:
    """The path that meets the north star's <= 1e-3 tolerance is the fp32 engine (exact-fp32 MFMA GEMMs, `precision: 32-true`);
    the headline number is the bf16 engine.  This record says what the 1e-3 path costs: ms / step of the SAME training step on
    the fp32 engine at a batch that fits, and — computed in this run — its forward error against the oracle's fp32 forward
    (max |y - y_ref| / max |y_ref| on the oracle child's fixture: same weights, same input)."""
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.step import TrainStep
    from viscy_amd.unext2 import UNeXt2

    torch.manual_seed(42)
    model = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True,
                   head_expansion_ratio=4, decoder_conv_blocks=2).to(dev)
    rec = {"dtype": "fp32", "per_gpu_batch": B, "tolerance": 1e-3}
    if fixture_path and os.path.exists(fixture_path):
        fx = torch.load(fixture_path, map_location="cpu")
        model.load_state_dict(fx["state_dict"], strict=True)
        model.compute_dtype, model.grad_mode = torch.float32, "autograd"
        with torch.no_grad():
            y = model(fx["x"].to(dev)).float().cpu()
        err = ((y - fx["y"]).abs().max() / fx["y"].abs().max()).item()
        rec["forward_rel_err_vs_oracle"] = float(f"{err:.3e}")
        rec["forward_within_tolerance"] = bool(err <= 1e-3)
        rec["checked_on"] = f"oracle fp32 forward, B={fx['x'].shape[0]}, Z=5, 256x256, same weights and input"
        model._engine = None
    else:
        rec["forward_rel_err_vs_oracle"] = None
        rec["checked_on"] = "the oracle child did not deliver its fixture in time"
    nonzero_grn_(model)
    model.compute_dtype, model.grad_mode = torch.float32, "flat"
    opt = FlatAdamW(model.engine(), lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=steps + 4, warmup_multiplier=1e-3)
    x, tgt = make_batch(B, 256, 256, dev, seed=11)
    step = TrainStep(model, MixedLoss(0.5, 0.0, 0.5), opt, None, use_graph=False)
    step(x, tgt)
    step(x, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x, tgt)
    torch.cuda.synchronize()
    dt_s = (time.perf_counter() - t0) / steps
    rec.update({"ms_per_step": round(dt_s * 1e3, 3), "patches_per_s": round(B / dt_s, 1), "steps": steps,
                "loss_finite": bool(torch.isfinite(loss).item())})
    return rec


class Watchdog:
    """A phase of a multi-rank run that does not finish in `seconds` ends the process with a message that names the rank and
    the phase (exit code 17) instead of hanging the launcher: a collective whose peer never arrives has no other way out."""

    def __init__(self, phase: str, seconds: float, rank: int):
        import threading

        self.phase, self.seconds, self.rank = phase, seconds, rank
        self.timer = threading.Timer(seconds, self._expire)
        self.timer.daemon = True

    def _expire(self):
        sys.stderr.write(f"bench.py: rank {self.rank} did not finish '{self.phase}' within {self.seconds:.0f} s — giving up (exit 17)\n")
        sys.stderr.flush()
        os._exit(17)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()


def ranks_agree(flat: torch.Tensor) -> tuple[bool, float]:
    """every rank holds the same parameters: an order-independent checksum of the flat fp32 buffer (sum of its bit patterns as
    int64, and the float64 sum) is all-reduced with MIN and MAX — identical on every rank iff the two agree.  Backend-agnostic
    (RCCL in the bench, gloo in tests/test_bench_cpu.py)."""
    import torch.distributed as dist

    bits = flat.detach().view(torch.int32).to(torch.int64).sum().reshape(1)
    fsum = flat.detach().double().sum().reshape(1)
    lo = torch.cat([bits.double(), fsum])
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi)), float(fsum.item())


def rank_times(elapsed_s: float, steps: int, device) -> dict:
    """per-rank ms / step of the timed region, gathered on every rank: min, max and the list"""
    import torch.distributed as dist

    t = torch.tensor([elapsed_s * 1e3 / steps], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    v = [round(o.item(), 3) for o in out]
    return {"min": min(v), "max": max(v), "per_rank": v}


def launch_command(n: int, argv: list[str], port: int | None = None) -> list[str]:
    """the command `bench.py --gpus N` re-executes itself under when it is started as ONE process (the driver's form for
    N = 1; for N > 1 the driver starts torch.distributed.run itself and WORLD_SIZE is set): one rank per GPU of one node,
    rendezvous on 127.0.0.1 (the reference's topology: recipes/topology/ddp_4gpu.yml:2-6 — `strategy: ddp`, `devices: 4`)"""
    if port is None:
        port = 29500 + os.getpid() % 2000
    rest = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + rest


def resolve_world(gpus: int, env: dict, visible_devices: int) -> tuple[str, int]:
    """What a `bench.py --gpus N` process has to do, from N, the launcher environment and the number of visible GPUs:
    ("run", world) — this process is a rank (or the only one); ("spawn", N) — started alone with N > 1: re-execute under
    torch.distributed.run; raises SystemExit (non-zero) when the request cannot be honoured — a line with n_gpus != N is never
    printed."""
    if gpus < 1:
        raise SystemExit(f"bench.py: --gpus {gpus} is not a GPU count")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to "
                             f"print a line whose n_gpus is not what was asked for")
        if visible_devices < world:
            raise SystemExit(f"bench.py: {world} ranks but only {visible_devices} GPU(s) visible on this node")
        return "run", world
    if gpus == 1:
        return "run", 1
    if visible_devices < gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} needs {gpus} visible GPUs on this node, found {visible_devices}")
    return "spawn", gpus


def main():
    if "--cpu-baseline-child" in sys.argv:
        i = sys.argv.index("--cpu-baseline-child")
        fx = sys.argv[i + 1] if i + 1 < len(sys.argv) else None
        nt = int(sys.argv[sys.argv.index("--threads") + 1]) if "--threads" in sys.argv else None
        _cpu_baseline_child(None if fx == "-" else fx, nt)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node to run on: one rank per GPU.  Started as a single process "
                    "with N > 1, bench.py re-executes itself under torch.distributed.run with N ranks")
    ap.add_argument("--dry-launch", action="store_true", help="print the N-rank launch command as JSON and exit (no GPU needed)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("VSX_BENCH_BATCH", 512)),
                    help="patches per GPU per step (weak scaling: the same value at --gpus 1 / 2 / 4 / 8; 512 needs ~130 GB of the 288 GB "
                         "per GPU and is the pixel count of the north star's B = 8 x 2048^2 stack; 128 / 256 run at 57 / 63 %% of its rate)")
    ap.add_argument("--no-fp32-parity", action="store_true", help="skip the fp32-engine sub-record (ms / step of the <= 1e-3 path + its "
                    "forward error against the oracle)")
    ap.add_argument("--collective-timeout", type=float, default=180.0, help="seconds a rank waits in process-group initialisation, in "
                    "the first all-reduce and in the first training step before it gives up and names itself (N > 1)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gate", action="store_true", help="skip the (B=8, 2048^2) gate-shape sub-record")
    ap.add_argument("--gate-steps", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured step")
    ap.add_argument("--profile-ops", action="store_true", help="print the per-op-class event-timing table to stderr")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel collective path even with one rank (RCCL smoke test on a single GPU)")
    args = ap.parse_args()

    if args.dry_launch:
        print(json.dumps({"dry_launch": True, "n_gpus": args.gpus,
                          "command": launch_command(args.gpus, sys.argv[1:]) if args.gpus > 1 else [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--dry-launch"]}))
        return
    action, world = resolve_world(args.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0)
    if action == "spawn":
        import subprocess

        # rank 0 of the child job prints the one JSON line on the inherited stdout; this process only forwards the exit code
        raise SystemExit(subprocess.call(launch_command(world, sys.argv[1:])))

    # stdout carries exactly ONE line, the JSON result: everything any library writes to file descriptor 1 during the run
    # (RCCL prints a version banner through C stdio at communicator init and flushes it at exit, i.e. AFTER a Python print)
    # is routed to stderr, and the result is written to the saved descriptor at the very end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dp
    if multi:
        import datetime

        import torch.distributed as dist

        # No NCCL_* / RCCL_* variable is set or relied on (HSA_ENABLE_IPC_MODE_LEGACY=0 comes from the image: dmabuf IPC).  A rank
        # whose peer never arrives leaves through the watchdog with its rank and phase on stderr instead of hanging the launcher.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with Watchdog("init_process_group (RCCL communicator over xGMI)", args.collective_timeout + 30, rank):
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.collective_timeout))
        with Watchdog("first all-reduce", args.collective_timeout, rank):
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != int(os.environ["WORLD_SIZE"]):
                raise SystemExit(f"bench.py: rank {rank}: the first all-reduce summed to {probe.item()} over {os.environ['WORLD_SIZE']} ranks")

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
    with Watchdog("first training step", args.collective_timeout + 120, rank):
        if not args.profile_ops:
            eager(x, tgt)  # first launches load code objects / size workspaces: keep that out of the instrumented step
        with OpTimer(ops) as tm:  # one eager, instrumented step: per-op-class times → dominant kernel class
            l0 = eager(x, tgt)
        table = tm.summary()
    agree = None
    if multi:
        # data parallel means identical parameters on every rank after every step: checked once, after the first two steps
        # (different batches per rank, all-reduced gradients, AdamW) — a rank that diverged makes the whole line meaningless
        with Watchdog("parameter checksum all-reduce", args.collective_timeout, rank):
            agree, _ = ranks_agree(eng.flat)
        if not agree:
            raise SystemExit(f"bench.py: rank {rank}: parameters differ between ranks after the first optimisation steps")
    for _ in range(max(args.warmup, 1)):
        step()
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
    REPLAYS = 3
    with OpTimer(ops) as tm:
        for _ in range(REPLAYS):
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
    per_rank = None
    if multi:
        per_rank = rank_times(elapsed, args.steps, dev)
        elapsed = per_rank["max"] * args.steps / 1e3  # the contract: MAX over ranks
    fam_table = tm.summary()
    tmpl_table = tm.summary(templates=True)
    # the dominant KERNEL: the template instantiation (kernel family + the template arguments / GEMM shape that select the code)
    # with the largest summed launch time of the step; `roofline_classes` keeps the family view next to it
    dominant = max(tmpl_table, key=lambda c: tmpl_table[c]["ms"]) if tmpl_table else None
    dom = tmpl_table.get(dominant) if dominant else None

    binfo = build_info()
    cpu_leg = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_leg = CpuBaseline()  # writes the fp32 parity fixture now; its timing phase is released in result(), after the GPU sub-records
    gate, peak_main, parity = None, None, None
    sub = rank == 0 and world == 1 and args.size == 256 and args.dtype == "bf16"
    want_gate, want_parity = sub and not args.no_gate, sub and not args.no_fp32_parity
    if want_gate or want_parity:
        # Sub-records measured in the same driver run.  The headline model / graph / batch are released first (the headline and
        # the gate shape each need ~130 GB of the 288 GB).
        loss_keep = loss.detach().clone()
        peak_main = torch.cuda.max_memory_reserved()
        del graphed, eager, infer, tm, loss, step
        model._engine = None
        opt = ddp = eng = None
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        if want_gate:  # the north star's roofline gate shape: (B = 8, Z = 5, 2048 x 2048) training step
            gate = gate_shape_record(dev, steps=args.gate_steps)
            gc.collect()
            torch.cuda.empty_cache()
        if want_parity:  # the fp32 engine (the <= 1e-3 path): its step time and its forward error against the oracle
            parity = fp32_parity_record(dev, cpu_leg.fixture() if cpu_leg else None)
        loss = loss_keep
    if rank == 0:
        patches = world * B * args.steps
        value = patches / elapsed
        scale = (args.size / 256.0) ** 2
        roof, roof_classes = None, []
        # HBM traffic per kernel family: PMC counters cannot be read from inside this process, so the figures come from the
        # committed rocprofv3 --pmc passes of the SAME configuration (scripts/pmc_traffic.sh -> profiles/<PROFILE_ROUND>_pmc_traffic_b<batch>.json:
        # FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on a launch with a known byte count as MI355X_MICROARCH.md
        # prescribes).  The file records the kernel-source hash and the flag set it was measured with; a file from other
        # sources / flags is refused (traffic stays null).
        tfam, tfam_k, trefused, tf_path = {}, {}, None, os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_traffic_b{B}.json")
        if args.size == 256 and args.dtype == "bf16" and os.path.exists(tf_path):
            try:
                tfj = json.load(open(tf_path))
                if tfj.get("source_hash") == binfo["source_hash"] and tfj.get("flags") == binfo["flags"]:
                    tfam = tfj.get("families", {})
                    tfam_k = tfj.get("templates", {})
                else:
                    trefused = "profiles file was measured on other kernel sources / flags"
            except (OSError, ValueError, KeyError):
                pass

        def family_record(fam, d):
            """one kernel family over the REPLAYS instrumented eager steps behind the timed region"""
            ms_step = d["ms"] / REPLAYS
            gb = d["bytes"] / REPLAYS / 1e9
            rec = {"family": fam, "ms_per_step": round(ms_step, 3), "launches_per_step": round(d["launches"] / REPLAYS, 1),
                   "algorithmic_GB_per_step": round(gb, 3),
                   "hbm_frac": round(gb / (ms_step * 1e-3) / HBM_PEAK_GBS, 4),
                   "tflops": round(d["flops"] / REPLAYS / (ms_step * 1e-3) / 1e12, 1),
                   "mfma_frac": round(d["flops"] / REPLAYS / (ms_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
            # priced both ways: design bytes (what the launches are built to move, incl. the 4C-wide g / dh and the per-sample
            # fp32 products) and the bytes SURVEY §8(d)'s floor contains for them (layer-boundary tensors only)
            fgb = d["floor_bytes"] / REPLAYS / 1e9
            rec["floor_GB_per_step"] = round(fgb, 3)
            rec["hbm_frac_floor_bytes"] = round(fgb / (ms_step * 1e-3) / HBM_PEAK_GBS, 4)
            t = tfam.get(fam)
            if t:
                rec["traffic_GB_per_step"] = round(t["read_GB_per_step"] + t["write_GB_per_step"], 3)
                rec["traffic_over_algorithmic"] = round(rec["traffic_GB_per_step"] / max(gb, 1e-9), 3)
            else:
                rec["traffic_GB_per_step"] = None
            return rec

        for fam, d in sorted(fam_table.items(), key=lambda kv: -kv[1]["ms"])[:6]:
            roof_classes.append(family_record(fam, d))
        if dom:
            ms = dom["ms"] / dom["launches"]
            tf = dom["flops"] / dom["launches"] / (ms * 1e-3) / 1e12
            gbs = dom["bytes"] / dom["launches"] / (ms * 1e-3) / 1e9
            if tf / MFMA_BF16_PEAK_TFLOPS >= gbs / HBM_PEAK_GBS:
                roof = {"kernel": dominant, "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                        "avg_launch_ms": round(ms, 5), "launches": dom["launches"],
                        "hbm_side_GBs": round(gbs, 1)}
            else:
                roof = {"kernel": dominant, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(ms, 5),
                        "launches": dom["launches"], "mfma_frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4)}
            roof["family"] = dom["family"]
            roof["launches_per_step"] = round(dom["launches"] / REPLAYS, 1)
            roof["ms_per_step"] = round(dom["ms"] / REPLAYS, 3)
            roof["strict_frac"] = round(dom["strict_bytes"] / dom["launches"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)  # operands + ONE output only
            roof["algorithmic_bytes_per_launch"] = round(dom["bytes"] / dom["launches"])
            roof["frac_floor_bytes"] = round(dom["floor_bytes"] / dom["launches"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            # PMC traffic of this template from the committed --pmc passes (per kernel name: tools/pmc_traffic.py `kernels`)
            t = (tfam_k or {}).get(dominant)
            if t:
                roof["traffic"] = round((t["read_GB_per_step"] + t["write_GB_per_step"]) * 1e9 / max(t["launches_per_step"], 1e-9))
                roof["traffic_source"] = os.path.relpath(tf_path, ROOT)
            elif trefused:
                roof["traffic_refused"] = trefused
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
            "roofline_classes": roof_classes,
            "gate_shape": gate,
            "fp32_parity": parity,
            "build": binfo,
            "fwd": {"ms_per_pass": round(fwd_s * 1e3, 3), "patches_per_s_per_gpu": round(B / fwd_s, 1),
                    "algorithmic_hbm_GBps": round(B / fwd_s * FWD_MB_PER_PATCH * scale / 1e3, 1),
                    "frac_hbm_peak": round(B / fwd_s * FWD_MB_PER_PATCH * scale * 1e6 / (HBM_PEAK_GBS * 1e9), 4),
                    "tflops": round(B / fwd_s * FWD_GFLOP_PER_PATCH * scale / 1e3, 1)},
        }
        if not res["whole_path"]["loss_finite"]:
            print("bench.py: WARNING: the training loss is not finite after the timed steps — this measurement is INVALID",
                  file=sys.stderr)
        if multi:
            res["ranks"] = {"ms_per_step": per_rank, "parameters_identical_across_ranks": agree}
        if cpu_leg is not None:  # the CPU leg is reported at N = 1 only (the other ranks would idle)
            res["cpu_baseline"] = cpu_leg.result()
    if multi:
        with Watchdog("final barrier", args.collective_timeout, rank):
            torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    os.close(result_fd)


if __name__ == "__main__":
    main()
