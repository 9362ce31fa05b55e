"""GPU: long-run and repeat-launch robustness of the kernel family.

* read-before-write: a whole training step with every ``torch.empty`` allocation filled with NaN stays finite;
* repeat-launch determinism: each lean NT / TN / LayerNorm / GRN-GELU / depthwise instantiation launched hundreds of
  times on fixed inputs — outputs a kernel owns (no atomics) must be BIT-identical to the first launch, reduction
  outputs (fp32 atomics, order varies) identical to round-off;
* soak: >= 300 hipGraph-replayed bf16 training steps at B = 128 with a finiteness check of loss, gradients and
  parameters after every step, with the shipped flags and with the streaming-access flags on.
"""

import os
import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

REPEAT = int(os.environ.get("VSX_SOAK_REPEAT", 300))
SOAK_STEPS = int(os.environ.get("VSX_SOAK_STEPS", 300))


def _bench_model(dt=torch.bfloat16, backbone="convnextv2_tiny", seed=42):
    import bench
    from viscy_amd.unext2 import UNeXt2

    torch.manual_seed(seed)
    m = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone=backbone, head_pool=True, head_expansion_ratio=4,
               decoder_conv_blocks=2).cuda()
    bench.nonzero_grn_(m)
    m.compute_dtype, m.grad_mode = dt, "flat"
    return m


@pytest.mark.parametrize("dt,backbone,B,S", [(torch.bfloat16, "convnextv2_tiny", 3, 256), (torch.float32, "convnextv2_femto", 2, 192),
                                             (torch.bfloat16, "convnextv2_tiny", 1, 192)],
                         ids=["bf16_tiny_256", "f32_femto_192", "bf16_tiny_192"])
def test_training_step_with_poisoned_allocations_is_finite(dt, backbone, B, S):
    """every buffer the schedule allocates with torch.empty is NaN-filled: any element a kernel reads before some kernel
    wrote it (partial tiles, tails, skipped rows) surfaces as a NaN in the loss / gradients, deterministically"""
    import bench
    from viscy_amd import debug, ops
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.step import TrainStep

    m = _bench_model(dt, backbone)
    eng = m.engine()
    opt = FlatAdamW(eng, lr=2e-4)
    x, t = bench.make_batch(B, S, S, "cuda")
    step = TrainStep(m, MixedLoss(0.5, 0, 0.5), opt, use_graph=False)
    with debug.poison_empty():
        for _ in range(2):
            with debug.FiniteGuard(ops) as g:
                loss = step(x, t)
            assert g.first is None
            assert torch.isfinite(loss).item()
            assert torch.isfinite(eng.flat_grad).all().item() and torch.isfinite(eng.flat).all().item()
        m.eval()
        with torch.no_grad(), debug.FiniteGuard(ops):
            y = m(x)
        assert torch.isfinite(y).all().item()


# ------------------------------------------------------------------ repeat-launch determinism
def _rnd(*shape, dt=torch.bfloat16, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt).cuda()


def _repeat(launch, exact, approx=(), n=None, rtol=2e-4):
    """launch() -> dict name -> tensor; `exact` names must be bit-identical across launches, `approx` within round-off"""
    n = n or REPEAT
    first = {k: v.clone() for k, v in launch().items()}
    for k, v in first.items():
        assert torch.isfinite(v.float()).all().item(), f"{k}: non-finite on the first launch"
    for i in range(n):
        out = launch()
        for k in exact:
            if not torch.equal(out[k], first[k]):
                d = (out[k].float() - first[k].float())
                nbad = int((d != 0).sum().item()) + int((~torch.isfinite(out[k].float())).sum().item())
                raise AssertionError(f"launch {i + 1}: output '{k}' differs from the first launch in {nbad} elements "
                                     f"(max |diff| {d.abs().nan_to_num(float('inf')).max().item():.3e})")
        for k in approx:
            a, b = out[k].float(), first[k].float()
            assert torch.isfinite(a).all().item(), f"launch {i + 1}: reduction '{k}' non-finite"
            err = (a - b).abs().max().item() / b.abs().max().clamp_min(1e-20).item()
            assert err <= rtol, f"launch {i + 1}: reduction '{k}' moved by {err:.2e}"


NT_CASES = [
    # M, N, K, hw, epi, pro  — the lean instantiations that carry the step (BK = 32 / 64, every epilogue kind, GRN prologue)
    (16384, 384, 96, 4096, "gelu_sq", False),
    (16384, 896, 224, 4096, "gelu_sq", False),
    (4096, 1536, 384, 256, "gelu_sq", False),
    (16384, 96, 384, 4096, "res", False),
    (4096, 384, 1536, 256, "res", True),
    (2048, 768, 3072, 128, "res", True),
    (16384, 384, 96, 4096, "dz", False),
    (4096, 1536, 384, 256, "dz", False),
    (16384, 96, 384, 4096, "none", False),
    (8192, 224, 896, 4096, "none", False),
    (8192, 192, 224, 1024, "bias", False),
]


@pytest.mark.parametrize("stream_flags", [0, 3], ids=["nt_stream0", "nt_stream3"])
@pytest.mark.parametrize("M,N,K,hw,epi,pro", NT_CASES)
def test_lean_nt_gemm_is_deterministic(M, N, K, hw, epi, pro, stream_flags):
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    A, Bw = _rnd(M, K, seed=1), _rnd(N, K, seed=2, scale=0.05)
    bias = _rnd(N, dt=torch.float32, seed=3)
    nb = M // hw
    saved_flag = L.lib().vsx_get_flag(b"nt_stream")
    L.lib().vsx_set_flag(b"nt_stream", stream_flags)
    try:
        if epi == "gelu_sq":
            C, C2 = torch.empty((M, N), dtype=dt, device="cuda"), torch.empty((M, N), dtype=dt, device="cuda")

            def launch():
                C.fill_(float("nan")); C2.fill_(float("nan"))
                red = torch.zeros((nb, N), dtype=torch.float32, device="cuda")
                ops.gemm("nt", A, Bw, C, M, N, K, K, K, N, dtype=dt, epi=L.EPI_BIAS_GELU_SQ, bias=bias, red0=red, hw=hw, C2=C2)
                return {"h": C, "g": C2, "colsq": red}

            _repeat(launch, ["h", "g"], ["colsq"])
        elif epi == "res":
            C = torch.empty((M, N), dtype=dt, device="cuda")
            res = _rnd(M, N, seed=4)
            s = (1 + 0.1 * _rnd(nb, K, dt=torch.float32, seed=5)) if pro else None
            gb = 0.1 * _rnd(K, dt=torch.float32, seed=6) if pro else None

            def launch():
                C.fill_(float("nan"))
                ops.gemm("nt", A, Bw, C, M, N, K, K, K, N, dtype=dt, epi=L.EPI_BIAS_RES, bias=bias, res=res, ldr=N, hw=hw,
                         pro=L.PRO_GRN if pro else L.PRO_NONE, grn_s=s, grn_b=gb)
                return {"out": C}

            _repeat(launch, ["out"])
        elif epi == "dz":
            C = torch.empty((M, N), dtype=dt, device="cuda")
            aux = _rnd(M, N, seed=7)

            def launch():
                C.fill_(float("nan"))
                red = torch.zeros((2, nb, N), dtype=torch.float32, device="cuda")
                ops.gemm("nt", A, Bw, C, M, N, K, K, K, N, dtype=dt, epi=L.EPI_DZ, aux=aux, ldx=N, red0=red[0], red1=red[1], hw=hw)
                return {"dz": C, "PS": red}

            _repeat(launch, ["dz"], ["PS"], rtol=2e-3)
        else:
            C = torch.empty((M, N), dtype=dt, device="cuda")

            def launch():
                C.fill_(float("nan"))
                ops.gemm("nt", A, Bw, C, M, N, K, K, K, N, dtype=dt, epi=L.EPI_BIAS if epi == "bias" else L.EPI_NONE,
                         bias=bias if epi == "bias" else None)
                return {"out": C}

            _repeat(launch, ["out"])
    finally:
        L.lib().vsx_set_flag(b"nt_stream", saved_flag)


@pytest.mark.parametrize("M,N,K,hw,pro", [(16384, 96, 384, 4096, True), (16384, 384, 96, 4096, False), (16384, 224, 896, 4096, True),
                                          (16384, 896, 224, 4096, False), (8192, 384, 1536, 256, True), (8192, 768, 3072, 64, True),
                                          (8192, 768, 192, 1024, False)])
def test_lean_tn_gemm_is_stable(M, N, K, hw, pro):
    """weight-gradient GEMMs accumulate with fp32 atomics: results must agree to round-off and stay finite"""
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dt = torch.bfloat16
    Y, X = _rnd(M, K, seed=1), _rnd(M, N, seed=2, scale=0.05)
    nb = M // hw
    s = (1 + 0.1 * _rnd(nb, K, dt=torch.float32, seed=5)) if pro else None
    gb = 0.1 * _rnd(K, dt=torch.float32, seed=6) if pro else None

    def launch():
        W = torch.zeros((N, K), dtype=torch.float32, device="cuda")
        cs = torch.zeros(N, dtype=torch.float32, device="cuda")
        ops.gemm("tn", Y, X, W, M, N, K, K, N, K, dtype=dt, pro=L.PRO_GRN if pro else L.PRO_NONE, grn_s=s, grn_b=gb, hw=hw, colsum=cs)
        return {"W": W, "cs": cs}

    _repeat(launch, [], ["W", "cs"], n=max(REPEAT // 3, 20), rtol=1e-3)


@pytest.mark.parametrize("stream_flags", [0, 3], ids=["ln_stream0", "ln_stream3"])
@pytest.mark.parametrize("rows,C", [(65536, 96), (65536, 224), (16384, 384), (4096, 768), (16384, 576)])
def test_layernorm_is_deterministic(rows, C, stream_flags):
    from viscy_amd import _lib as L
    from viscy_amd import debug, ops

    x, dy = _rnd(rows, C, seed=1), _rnd(rows, C, seed=2)
    add = _rnd(rows, C, seed=3)
    gam, bet = _rnd(C, dt=torch.float32, seed=4), _rnd(C, dt=torch.float32, seed=5)
    saved_flag = L.lib().vsx_get_flag(b"ln_stream")
    L.lib().vsx_set_flag(b"ln_stream", stream_flags)
    try:
        def fwd():
            y, mean, rstd = ops.ln_fwd(x, None, None, rows, C, need_mean=False)
            y2, mean2, rstd2 = ops.ln_fwd(x, gam, bet, rows, C)
            return {"y": y, "rstd": rstd, "y2": y2, "mean2": mean2, "rstd2": rstd2}

        with debug.poison_empty():
            _repeat(fwd, ["y", "rstd", "y2", "mean2", "rstd2"], n=max(REPEAT // 3, 20))
            f = fwd()
            xh, rstd, mean2, rstd2 = f["y"].clone(), f["rstd"].clone(), f["mean2"].clone(), f["rstd2"].clone()

            def bwd():
                dx = ops.ln_bwd(dy, xh, None, rstd, None, None, None, None, rows, C)
                dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
                dx2 = ops.ln_bwd(dy, x, mean2, rstd2, gam, add, dg, db, rows, C)
                return {"dx": dx, "dx2": dx2, "dg": dg, "db": db}

            _repeat(bwd, ["dx", "dx2"], ["dg", "db"], n=max(REPEAT // 3, 20), rtol=1e-3)
    finally:
        L.lib().vsx_set_flag(b"ln_stream", saved_flag)


@pytest.mark.parametrize("stream_flags", [0, 2], ids=["grn_stream0", "grn_stream2"])
@pytest.mark.parametrize("M,N,hw", [(16384, 384, 4096), (16384, 896, 4096), (4096, 1536, 256), (2048, 3072, 64)])
def test_grn_gelu_bwd_is_deterministic(M, N, hw, stream_flags):
    from viscy_amd import _lib as L
    from viscy_amd import ops

    dz0, h = _rnd(M, N, seed=1), _rnd(M, N, seed=2)
    nb = M // hw
    s, t = 1 + 0.1 * _rnd(nb, N, dt=torch.float32, seed=3), 0.01 * _rnd(nb, N, dt=torch.float32, seed=4)
    saved_flag = L.lib().vsx_get_flag(b"grn_stream")
    L.lib().vsx_set_flag(b"grn_stream", stream_flags)
    try:
        def launch():
            dz = dz0.clone()
            cs = torch.zeros(N, dtype=torch.float32, device="cuda")
            ops.grn_gelu_bwd(dz, h, s, t, cs, M, N, hw)
            return {"dh": dz, "cs": cs}

        _repeat(launch, ["dh"], ["cs"], n=max(REPEAT // 3, 20), rtol=1e-3)
    finally:
        L.lib().vsx_set_flag(b"grn_stream", saved_flag)


@pytest.mark.parametrize("B,H,W,C", [(4, 64, 64, 96), (4, 64, 64, 224), (8, 32, 32, 192), (16, 16, 16, 384), (32, 8, 8, 768)])
def test_dwconv7_is_deterministic(B, H, W, C):
    from viscy_amd import debug, ops

    x, dy = _rnd(B * H * W, C, seed=1), _rnd(B * H * W, C, seed=2)
    w, b = _rnd(49, C, dt=torch.float32, seed=3, scale=0.1), _rnd(C, dt=torch.float32, seed=4)

    def launch():
        y = ops.dwconv7_fwd(x, w, b, B, H, W, C)
        dx = ops.dwconv7_bwd_data(dy, w, x, B, H, W, C)
        dw, db = torch.zeros((49, C), device="cuda"), torch.zeros(C, device="cuda")
        ops.dwconv7_bwd_weight(dy, x, dw, db, B, H, W, C)
        return {"y": y, "dx": dx, "dw": dw, "db": db}

    with debug.poison_empty():
        _repeat(launch, ["y", "dx"], ["dw", "db"], n=max(REPEAT // 3, 20), rtol=1e-3)


# ------------------------------------------------------------------ soak
@pytest.mark.parametrize("flags", ["", "nt_stream=0,grn_stream=0,ln_stream=0"], ids=["default_flags_streaming_on", "streaming_off"])
def test_soak_graph_replayed_bf16_training(flags):
    """>= 300 hipGraph replays of the whole bf16 training step at B = 128: loss, every gradient and every parameter stay
    finite after every step, and the loss of the (fixed) batch ends below where it started"""
    import bench
    from viscy_amd import _lib as L
    from viscy_amd import debug
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.step import TrainStep

    lib = L.lib()
    saved = {}
    for kv in filter(None, flags.split(",")):
        k, _, v = kv.partition("=")
        saved[k] = lib.vsx_get_flag(k.encode())
        lib.vsx_set_flag(k.encode(), int(v))
    try:
        m = _bench_model()
        eng = m.engine()
        opt = FlatAdamW(eng, lr=2e-4, schedule="WarmupCosine", warmup_steps=3, t_total=SOAK_STEPS, warmup_multiplier=1e-3)
        x, t = bench.make_batch(128, 256, 256, "cuda")
        step = TrainStep(m, MixedLoss(0.5, 0, 0.5), opt, use_graph=True)
        res = debug.soak(lambda: step(x, t), SOAK_STEPS, [eng.flat, opt.m, opt.v],
                         lambda: {"grad": eng.flat_grad, "param": eng.flat})
        assert res["first_bad_step"] is None, f"non-finite {res['what']} at step {res['first_bad_step']} (losses {res['losses'][-5:]})"
        assert res["losses"][-1] < res["losses"][0]
    finally:
        for k, v in saved.items():
            lib.vsx_set_flag(k.encode(), v)


# ------------------------------------------------------------------ optimiser state lives on the device (pinned-hyper race)
def test_unsynced_graph_replays_follow_the_warmup_cosine_schedule():
    """N hipGraph replays enqueued WITHOUT any host synchronisation (the host runs far ahead of the device, as in bench.py)
    must apply lr(0), lr(1), ... in order with Adam's bias corrections of steps 1, 2, ...: on a gradient that does not depend
    on the parameters (the captured body only re-fills the gradient buffer) the trajectory equals torch.optim.AdamW +
    LambdaLR(WarmupCosine) exactly.  A busy-kernel in front of the replays keeps the device behind the host."""
    from viscy_amd.optim import FlatAdamW, warmup_cosine_lambda

    class _Eng:
        pass

    n, steps = 1 << 20, 40
    torch.manual_seed(0)
    eng = _Eng()
    eng.flat = torch.randn(n, device="cuda")
    eng.flat_grad = torch.zeros(n, device="cuda")
    gsrc = torch.randn(n, device="cuda")
    p_ref = torch.nn.Parameter(eng.flat.detach().cpu().clone())
    topt = torch.optim.AdamW([p_ref], lr=3e-3)
    sch = torch.optim.lr_scheduler.LambdaLR(topt, lambda s: warmup_cosine_lambda(s, 3, steps, 1e-3))
    opt = FlatAdamW(eng, lr=3e-3, schedule="WarmupCosine", warmup_steps=3, t_total=steps, warmup_multiplier=1e-3)
    opt.host_prepare()
    opt.t -= 1
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.flat_grad.copy_(gsrc)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.flat_grad.copy_(gsrc)
        opt.device_step()
    big = torch.randn(8192, 8192, device="cuda")
    for _ in range(30):  # ~100+ ms of queued work: every replay below is enqueued before the first one runs
        big = big @ big * 1e-4
    for _ in range(steps):
        opt.host_prepare()
        g.replay()
    torch.cuda.synchronize()
    for _ in range(steps):
        p_ref.grad = gsrc.cpu().clone()
        topt.step()
        sch.step()
    torch.testing.assert_close(eng.flat.cpu(), p_ref.detach(), rtol=2e-5, atol=2e-6)
    assert int(opt.step_dev) == steps == opt.t


def test_segmented_capture_equals_single_graph_and_leaves_state_untouched():
    """the step captured as three bucket segments + AdamW (the data-parallel layout, forced on one GPU) == the single
    graph == eager: same loss and gradients at lr = 0; capturing (two warm-up steps inside) changes neither parameters nor
    moments nor the step counter"""
    import bench
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.step import TrainStep

    x, t = bench.make_batch(2, 192, 192, "cuda")
    res = {}
    for mode in ("eager", "graph", "segments"):
        m = _bench_model(torch.float32, "convnextv2_femto", seed=0)
        eng = m.engine()
        opt = FlatAdamW(eng, lr=0.0, weight_decay=0.0)
        before = eng.flat.clone()
        step = TrainStep(m, MixedLoss(0.5, 0, 0.5), opt, use_graph=mode != "eager", segments=mode == "segments")
        losses = [float(step(x, t)) for _ in range(3)]
        if mode == "segments":
            assert len(step.graphs) == 3
        assert torch.equal(eng.flat, before) and int(opt.step_dev) == 3 and opt.t == 3
        assert float(opt.m.abs().max()) > 0  # moments did move (gradients are not zero)
        res[mode] = (losses, eng.flat_grad.clone())
    for mode in ("graph", "segments"):
        (le, ge), (lg, gg) = res["eager"], res[mode]
        assert max(abs(a - b) for a, b in zip(le, lg)) < 1e-5 * max(1.0, abs(le[0]))
        assert ((ge - gg).norm() / ge.norm()).item() < 1e-3
    # a real learning rate: the first captured step starts from the untouched initial state in every mode
    firsts = []
    for mode in ("eager", "graph", "segments"):
        m = _bench_model(torch.bfloat16, "convnextv2_tiny", seed=0)
        opt = FlatAdamW(m.engine(), lr=5e-4, schedule="WarmupCosine", warmup_steps=3, t_total=10, warmup_multiplier=1e-3)
        step = TrainStep(m, MixedLoss(0.5, 0, 0.5), opt, use_graph=mode != "eager", segments=mode == "segments")
        ls = [float(step(x, t)) for _ in range(8)]
        assert all(torch.isfinite(torch.tensor(ls))) and ls[-1] < ls[0]
        firsts.append(ls[0])
    assert max(firsts) - min(firsts) < 2e-3 * abs(firsts[0])


def test_capture_right_behind_a_collective_survives_the_rccl_watchdog():
    """Round 6: `bench.py --force-dp` died once in three runs with "operation not permitted when stream is capturing" raised from the
    ProcessGroupNCCL watchdog thread — it polls the events of collectives it has not retired yet (hipEventQuery), which a capture in
    the default global error mode forbids to EVERY thread.  The captures of viscy_amd.step run in thread-local mode; here a capture
    starts right behind asynchronous all-reduces, many times, while the watchdog still holds their work objects."""
    import torch.distributed as dist

    from viscy_amd.step import InferStep

    assert not dist.is_initialized()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29543")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        m = _bench_model(torch.bfloat16, "convnextv2_atto", seed=0).eval()
        x = torch.randn(1, 1, 5, 64, 64, device="cuda")
        buf = torch.ones(1 << 20, device="cuda")
        ref = None
        for i in range(12):
            works = [dist.all_reduce(buf, async_op=True) for _ in range(4)]
            step = InferStep(m)           # a fresh capture each time, the collectives' work objects still with the watchdog
            y = step(x).float().clone()
            for w in works:
                w.wait()
            if ref is None:
                ref = y
            assert torch.isfinite(y).all() and torch.allclose(y, ref, rtol=0, atol=2e-2 * float(ref.abs().max()))
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def test_rccl_one_rank_segments_with_interleaved_all_reduce():
    """RCCL at HEAD where the driver can see it: a 1-rank ``nccl`` process group, the step captured as
    three hipGraph segments with each bucket's ``all_reduce(async_op=True)`` issued between two replays (exactly what every
    rank of an N-GPU run executes), 24 steps — against the single-graph step without a process group: same losses, same
    parameters.  (No N > 1 scaling curve exists: the build session only reaches single-GPU boxes.)"""
    import torch.distributed as dist

    import bench
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.parallel import FlatDataParallel
    from viscy_amd.step import TrainStep

    x, t = bench.make_batch(4, 192, 192, "cuda")
    steps = 24

    def run(ddp_on):
        m = _bench_model(torch.bfloat16, "convnextv2_tiny", seed=0)
        eng = m.engine()
        opt = FlatAdamW(eng, lr=3e-4, schedule="WarmupCosine", warmup_steps=3, t_total=steps, warmup_multiplier=1e-3)
        ddp = FlatDataParallel(eng, opt, force=True) if ddp_on else None
        step = TrainStep(m, MixedLoss(0.5, 0, 0.5), opt, ddp, use_graph=True)
        if ddp_on:
            assert ddp.active and step.dist and step.segments
        losses = [float(step(x, t)) for _ in range(steps)]
        if ddp_on:
            assert len(step.graphs) == 3 and not ddp.works and not ddp._reduced   # every bucket reduced and waited for, each step
        torch.cuda.synchronize()
        return losses, eng.flat.clone()

    assert not dist.is_initialized()
    ref_losses, ref_flat = run(False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        losses, flat = run(True)
    finally:
        dist.destroy_process_group()
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    # same arithmetic up to the order of fp32 atomics inside the kernels — which this trajectory amplifies: the prediction of
    # a fresh network is uncorrelated with the target, so ms_ssim_25d's 1e-4 clamp on the per-sample contrast means sits at
    # the operating point for the first steps and round-off decides which (sample, scale) terms carry a gradient (see the
    # baseline-size fixture, oracle/validate_against_reference.py g8b).  Tight where the runs have not yet had a chance to
    # fork (observed agreement there: 1e-4), loose afterwards (observed up to 4.4e-2 at step 24)
    np.testing.assert_allclose(losses[:8], ref_losses[:8], rtol=2e-3)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-1)
    assert torch.nn.functional.cosine_similarity(flat - _bench_model(torch.bfloat16, "convnextv2_tiny", seed=0).engine().flat,
                                                 ref_flat - _bench_model(torch.bfloat16, "convnextv2_tiny", seed=0).engine().flat,
                                                 dim=0).item() > 0.9
