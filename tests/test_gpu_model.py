"""GPU: the full HIP path (UNeXt2 forward/backward, MixedLoss, AdamW) against the oracle
(oracle/*.py, pinned to the reference) and the committed golden fixtures.

Bar (BASELINE.json north_star): <= 1e-3 relative error in fp32 vs the reference CPU path;
bf16 production mode is checked with the tolerance the reference accepts for its own GPU
inference reproducibility test (atol 0.02 / rtol 1e-2, test_inference_reproducibility.py:70-73)."""

import pytest
import torch

from oracle import loss_ref, unext2_ref
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def _pair(kw, seed=7):
    from viscy_amd.unext2 import UNeXt2

    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=seed).eval()
    mine = UNeXt2(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    return ref, mine.cuda()


@pytest.mark.parametrize("tag", ["atto_pool", "femto_z15", "tiny_pool", "femto_preconv"])
def test_forward_matches_reference_golden_fp32(tag):
    """fixtures were produced by the REFERENCE's own unext2.py/blocks.py/heads.py (oracle/validate_against_reference.py G8)"""
    g = load_golden("unext2_forward.pt")[tag]
    _, mine = _pair(g["kwargs"], seed=g["seed"])
    mine.compute_dtype = torch.float32
    x = torch.randn(g["x_shape"], generator=torch.Generator().manual_seed(g["x_seed"]))
    with torch.no_grad():
        y = mine(x.cuda())
    assert y.shape == g["y"].shape and y.dtype == torch.float32
    assert relerr(y, g["y"]) <= 1e-3


CASES = [
    ("atto_pool", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_atto", head_pool=True), (2, 64, 96)),
    ("femto_z15", dict(in_channels=2, out_channels=2, in_stack_depth=15, out_stack_depth=5, backbone="convnextv2_femto"), (1, 64, 64)),
    ("tiny_pool", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True), (2, 128, 128)),
    # decoder_upsample_pre_conv=True (blocks.py:138-146): a dense 3x3 convolution in front of every decoder pixel shuffle
    ("femto_preconv", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_femto",
                           decoder_upsample_pre_conv=True), (2, 64, 96)),
]


@pytest.mark.parametrize("tag,kw,bhw", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("mode", ["autograd", "flat"])
@pytest.mark.parametrize("slope", ["slope1", "slope_ref"])
def test_forward_backward_vs_oracle_fp32(tag, kw, bhw, mode, slope):
    """Every parameter gradient of the whole model against the oracle's autograd.

    The head's PReLU derivative jumps at n̂ = 0.  With ~10^6 normalised values per batch about one lies within fp32
    rounding of the kink, and whether the HIP path and the CPU oracle put it on the same side depends on the last bit
    of the InstanceNorm mean (the statistics are reduced with atomics, i.e. in a run-dependent order).  One flipped
    voxel moves S1 = Σ dn for its channel, hence every upstream gradient, by a few 1e-3 — the comparison, not the
    kernel, is ill-conditioned there (measured in round 2 with a per-channel bisection of S1).  So the strict 2e-3 bar is applied with the
    slope set to 1 (no kink; dalpha still exercised), and with the reference's slope the bar is the direction of the
    full gradient plus a looser per-tensor bound."""
    torch.manual_seed(0)
    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=7).eval()
    if slope == "slope1":
        with torch.no_grad():
            ref.head.conv[0].adn.A.weight.fill_(1.0)
    from viscy_amd.unext2 import UNeXt2

    mine = UNeXt2(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    mine.compute_dtype = torch.float32
    mine.grad_mode = mode
    B, H, W = bhw
    x = torch.randn(B, kw["in_channels"], kw["in_stack_depth"], H, W)
    y = ref(x)
    dout = torch.randn_like(y)
    y.backward(dout)
    if mode == "flat":
        mine.engine().flat_grad.zero_()
    out = mine(x.cuda())
    assert relerr(out, y) <= 1e-3
    out.backward(dout.cuda())
    bar = 2e-3 if slope == "slope1" else 2e-2
    worst, worst_name = 0.0, ""
    gm, gr = [], []
    for (name, pr), pm in zip(ref.named_parameters(), mine.parameters()):
        assert pm.grad is not None, name
        if name == "head.conv.0.conv.bias":  # exactly-zero gradient (bias in front of InstanceNorm)
            assert pm.grad.abs().max().item() < 1e-3
            continue
        e = relerr(pm.grad, pr.grad)
        if e > worst:
            worst, worst_name = e, name
        assert e <= bar, (name, e)
        gm.append(pm.grad.flatten().cpu().double())
        gr.append(pr.grad.flatten().double())
    cos = torch.nn.functional.cosine_similarity(torch.cat(gm), torch.cat(gr), dim=0).item()
    assert 1 - cos < (1e-7 if slope == "slope1" else 2e-4), cos
    print(tag, mode, slope, "worst relative gradient error", worst, worst_name, "1-cos", 1 - cos)


def test_forward_backward_bf16_tracks_fp32():
    kw = CASES[2][1]
    ref, mine = _pair(kw)
    x = torch.randn(2, 1, 5, 128, 128, generator=torch.Generator().manual_seed(1))
    y = ref(x)
    dout = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(dout)
    gr = torch.cat([p.grad.flatten() for p in ref.parameters()]).double()
    # yardstick: the reference's OWN mixed-precision arithmetic — the same module under torch.autocast(bfloat16) (bf16
    # convolution / linear operands, fp32 LayerNorm statistics and residual sums), i.e. what Lightning's bf16-mixed runs —
    # measured against the same fp32 result.  The bf16 engine keeps the residual stream in bf16; the bar is that
    # this costs nothing measurable next to the rounding the reference's autocast already has.
    ref.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ya = ref(x)
    ya.float().backward(dout)
    ga = torch.cat([p.grad.flatten() for p in ref.parameters()]).double()
    ymax = y.detach().abs().max().item()
    ac_fwd = (ya.float().detach() - y.detach()).abs().max().item() / ymax
    ac_cos = torch.nn.functional.cosine_similarity(ga, gr, dim=0).item()
    mine.compute_dtype = torch.bfloat16
    out = mine(x.cuda())
    assert out.dtype == torch.float32
    hip_fwd = (out.cpu() - y.detach()).abs().max().item() / ymax
    out.backward(dout.cuda())
    gm = torch.cat([p.grad.flatten().cpu() for p in mine.parameters()]).double()
    cos = torch.nn.functional.cosine_similarity(gm, gr, dim=0).item()
    print(f"bf16 vs fp32 oracle: forward {hip_fwd:.4f} (reference autocast {ac_fwd:.4f}), gradient 1-cos {1 - cos:.2e} "
          f"(reference autocast {1 - ac_cos:.2e})")
    # measured (tools/bf16_gate.py, MI355X): forward 0.0107 vs 0.0112, 1-cos 1.70e-3 vs 1.88e-3
    assert hip_fwd <= max(1.25 * ac_fwd, 0.005) and hip_fwd <= 0.015, (hip_fwd, ac_fwd)
    assert 1 - cos <= 1.25 * (1 - ac_cos) and cos > 0.9975, (cos, ac_cos)
    # autocast contract: bf16 is selected by torch.autocast like Lightning's bf16-mixed
    mine.compute_dtype = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert mine._resolve_dtype() == torch.bfloat16
    assert mine._resolve_dtype() == torch.float32


def test_input_validation_and_state_dict_roundtrip():
    from viscy_amd.unext2 import UNeXt2

    m = UNeXt2(in_channels=1, out_channels=2, backbone="convnextv2_atto").cuda()
    with pytest.raises(ValueError, match="divisible by 32"):
        m(torch.zeros(1, 1, 5, 48, 64, device="cuda"))
    with pytest.raises(ValueError, match="expected input"):
        m(torch.zeros(1, 2, 5, 64, 64, device="cuda"))
    x = torch.randn(1, 1, 5, 64, 64, device="cuda")
    with torch.no_grad():
        y0 = m(x)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = UNeXt2(in_channels=1, out_channels=2, backbone="convnextv2_atto").cuda()
    m2.load_state_dict(sd, strict=True)
    with torch.no_grad():
        # reductions use fp32 atomics (order varies run to run) → agreement to round-off, not bitwise
        torch.testing.assert_close(m2(x), y0, rtol=1e-4, atol=1e-5)
    assert y0.shape == (1, 2, 5, 64, 64)


# ------------------------------------------------------------------ loss
@pytest.mark.parametrize("tag", ["rand_192", "corr_192", "corr_256", "corr_176x208"])
def test_mixed_loss_vs_reference_golden(tag):
    """golden values/gradients come from the reference's own ms_ssim_25d / MixedLoss (G2)."""
    from viscy_amd.losses import MixedLoss

    c = load_golden("loss.pt")[tag]
    gen = torch.Generator().manual_seed(c["seed"])
    target = torch.rand(c["shape"], generator=gen)
    pred = target + 0.1 * torch.randn(c["shape"], generator=gen) if c["corr"] else torch.rand(c["shape"], generator=gen)
    p = pred.cuda().requires_grad_(True)
    loss = MixedLoss(0.5, 0.0, 0.5)(p, target.cuda())
    assert abs(loss.item() - c["loss"].item()) <= 1e-3 * abs(c["loss"].item()), (loss.item(), c["loss"].item())
    loss.backward()
    g = p.grad.cpu()
    sample = g.flatten()[:: max(1, g.numel() // 4096)]
    scale = c["grad_absmax"].item()
    # per element against the reference's own gradient (the golden's strided sample).  Measured on MI355X
    # (tools/bf16_gate.py): worst element 3.4e-3 of the gradient's absolute maximum, mean |Δ| ≤ 2.2e-4 of mean |g|
    d = (sample - c["grad_sample"]).abs()
    assert d.max().item() <= 1e-2 * scale, (d.max().item(), scale)
    assert d.mean().item() <= 1e-3 * c["grad_sample"].abs().mean().item()
    assert abs(g.double().sum().item() - c["grad_sum"].item()) <= 2e-2 * max(abs(c["grad_sum"].item()), scale)
    # every element against the oracle run here (same seeds)
    pr = pred.clone().requires_grad_(True)
    lr = loss_ref.mixed_loss(pr, target, 0.5, 0.0, 0.5)
    lr.backward()
    dfull = (g - pr.grad).abs() / pr.grad.abs().max()
    assert dfull.max().item() <= 1e-2, dfull.max().item()
    assert (dfull > 5e-3).float().mean().item() <= 1e-3  # measured: the 99.9th percentile is ≤ 2.7e-3
    cos = torch.nn.functional.cosine_similarity(g.flatten().double(), pr.grad.flatten().double(), dim=0).item()
    assert cos > 0.99999, cos


@pytest.mark.parametrize("shape,a2", [((1, 2, 3, 181, 213), 0.0), ((2, 1, 5, 203, 177), 0.25), ((1, 1, 5, 224, 176), 0.0)])
def test_mixed_loss_one_pass_forward_odd_sizes(shape, a2):
    """the training forward is one pass per scale (SSIM sums + gradient field + the next scale's pooling, data range and the
    L1 / L2 sums: vsx_ssim_scale_fwd_fused); the forward without gradient still pools in a pass of its own.  Both against the
    oracle on odd plane sizes (a leftover row / column at every scale, tiles whose last column owns up to 42 pixels), a
    depth the kernel has no compile-time instance for, and a target that is an unaligned view."""
    from viscy_amd.losses import MixedLoss

    gen = torch.Generator().manual_seed(11)
    target = torch.rand(shape, generator=gen)
    pred = target + 0.1 * torch.randn(shape, generator=gen)
    flat = torch.empty(target.numel() + 1).cuda()
    flat[1:] = target.flatten().cuda()
    tdev = flat[1:].view(shape)  # 4-byte aligned only
    fn = MixedLoss(0.5, a2, 0.5)
    p = pred.cuda().requires_grad_(True)
    loss = fn(p, tdev)
    with torch.no_grad():
        loss_ng = fn(pred.cuda(), tdev)
    pr = pred.clone().requires_grad_(True)
    lr = loss_ref.mixed_loss(pr, target, 0.5, a2, 0.5)
    assert abs(loss.item() - lr.item()) <= 1e-3 * abs(lr.item()), (loss.item(), lr.item())
    assert abs(loss.item() - loss_ng.item()) <= 2e-6 * abs(lr.item()), (loss.item(), loss_ng.item())  # same math, other sum order
    loss.backward()
    lr.backward()
    g = p.grad.cpu()
    d = (g - pr.grad).abs() / pr.grad.abs().max()
    assert d.max().item() <= 1e-2, d.max().item()
    assert (d > 5e-3).float().mean().item() <= 1e-3
    cos = torch.nn.functional.cosine_similarity(g.flatten().double(), pr.grad.flatten().double(), dim=0).item()
    assert cos > 0.99999, cos


def test_mixed_loss_branches_and_errors():
    from viscy_amd.losses import MixedLoss

    gen = torch.Generator().manual_seed(3)
    t = torch.rand((2, 2, 5, 64, 64), generator=gen)
    p = torch.rand((2, 2, 5, 64, 64), generator=gen)
    # L1-only / L2-only branches need no 176-pixel minimum; bit-level agreement is not expected (sum order)
    for a1, a2 in [(1.0, 0.0), (0.0, 1.0), (0.3, 0.7)]:
        pc = p.cuda().requires_grad_(True)
        l = MixedLoss(a1, a2, 0.0)(pc, t.cuda())
        pr = p.clone().requires_grad_(True)
        lr = loss_ref.mixed_loss(pr, t, a1, a2, 0.0)
        assert abs(l.item() - lr.item()) <= 1e-5 * abs(lr.item())
        (l * 2.0).backward()
        (lr * 2.0).backward()
        assert relerr(pc.grad, pr.grad) <= 1e-4
    with pytest.raises(ValueError):
        MixedLoss(0, 0, 0)
    with pytest.raises(ValueError, match="176"):
        MixedLoss()(p.cuda(), t.cuda())
    with pytest.raises(RuntimeError, match="no CPU"):
        MixedLoss()(p, t)


# ------------------------------------------------------------------ one training step, end to end
def test_training_steps_reduce_loss_and_match_torch_adamw():
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.unext2 import UNeXt2

    torch.manual_seed(0)
    kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_femto", head_pool=True)
    m = UNeXt2(**kw).cuda()
    m.compute_dtype = torch.bfloat16
    m.grad_mode = "flat"
    eng = m.engine()
    opt = FlatAdamW(eng, lr=2e-3)
    crit = MixedLoss(0.5, 0.0, 0.5)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 1, 5, 192, 192), generator=g).cuda()
    tgt = (torch.nn.functional.avg_pool3d(x, (1, 5, 5), 1, (0, 2, 2)).repeat(1, 2, 1, 1, 1) * 0.5).contiguous()
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = crit(m(x), tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert all(l == l for l in losses) and min(losses[5:]) < losses[0], losses


# ------------------------------------------------------------------------------------------------ full-size properties
# The oracle needs minutes for one 256x256 patch batch; at BASELINE's full patch size the production (bf16) path is
# checked through properties that do not need it.
def _bench_model(seed=3):
    from viscy_amd.unext2 import UNeXt2

    torch.manual_seed(seed)
    m = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True).cuda()
    with torch.no_grad():  # GRN starts at zero in timm: give it something to do
        for n, p in m.named_parameters():
            if ".grn." in n:
                p.normal_(0.0, 0.2)
    m.compute_dtype = torch.bfloat16
    return m


def test_full_size_batch_independence_bf16():
    """every normalisation on the path (LayerNorm per pixel, GRN / InstanceNorm per sample) is batch-independent: a
    sample's output and its contribution to the gradient must not depend on what else is in the batch — this crosses
    every tile-inside-one-sample assumption of the lean kernels (per-sample weights, per-sample reductions) at the
    bench shape (256x256, Z=5), where the direct head convolution and the lean GEMM instantiations are the ones running."""
    m = _bench_model()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(6, 1, 5, 256, 256, generator=g).cuda()
    with torch.no_grad():
        yb = m(x)
        y1 = torch.cat([m(x[i : i + 1]) for i in (0, 3, 5)])
    assert yb.shape == (6, 2, 5, 256, 256)
    ref = yb[[0, 3, 5]]
    # same kernels, same per-sample arithmetic; only reduction orders (atomics) may differ
    assert ((ref - y1).abs().max() / ref.abs().max()).item() < 2e-2
    assert torch.nn.functional.cosine_similarity(ref.flatten(), y1.flatten(), dim=0).item() > 0.9999


def test_full_size_gradient_is_linear_in_dout_bf16():
    """backward is linear in the output gradient: grad(2·dout) = 2·grad(dout) up to bf16 rounding of the intermediates,
    and the gradient of a batch is the sum of the per-sample gradients (flat buffer, full bench patch size)."""
    m = _bench_model()
    m.grad_mode = "flat"
    eng = m.engine()
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 1, 5, 256, 256, generator=g).cuda()
    dout = torch.randn(2, 2, 5, 256, 256, generator=g).cuda()

    def grad_of(xx, dd):
        eng.flat_grad.zero_()
        m(xx).backward(dd)
        return eng.flat_grad.clone()

    g1 = grad_of(x, dout)
    g1b = grad_of(x, dout)
    g2 = grad_of(x, 2.0 * dout)
    # run-to-run floor: statistics are reduced with atomics, and a last-bit difference in a mean moves stored bf16
    # activations by one ulp here and there -> two identical calls agree to ~1e-3, not bit-wise
    floor = torch.nn.functional.cosine_similarity(g1, g1b, dim=0).item()
    cos = torch.nn.functional.cosine_similarity(g1, g2, dim=0).item()
    # measured: floor 0.9994 (gradients with heavy cancellation, e.g. the folded LayerNorm scales, carry the noise)
    assert floor > 0.998 and cos > floor - 2e-3, (floor, cos)
    assert abs((g2.norm() / g1.norm()).item() - 2.0) < 4e-2
    gs = grad_of(x[:1], dout[:1]) + grad_of(x[1:], dout[1:])
    assert torch.nn.functional.cosine_similarity(g1, gs, dim=0).item() > floor - 2e-3
    assert abs((gs.norm() / g1.norm()).item() - 1.0) < 4e-2


def test_full_size_mixed_loss_identities():
    """MixedLoss(t, t) = 0 with zero gradient for the MS-SSIM + L1 terms at full patch size; loss is symmetric under a
    permutation of the batch; the L1-only loss of (p, t) equals mean |p - t| computed by torch on the device."""
    from viscy_amd.losses import MixedLoss

    g = torch.Generator().manual_seed(13)
    t = torch.rand(4, 2, 5, 256, 256, generator=g).cuda()
    p = (t + 0.1 * torch.randn(t.shape, generator=g).cuda()).requires_grad_(True)
    crit = MixedLoss(0.5, 0.0, 0.5)
    same = t.clone().requires_grad_(True)
    l0 = crit(same, t)
    assert abs(l0.item()) < 1e-5
    l0.backward()
    assert same.grad.abs().max().item() < 1e-4
    la = crit(p, t)
    perm = torch.tensor([2, 0, 3, 1], device="cuda")
    lb = crit(p[perm], t[perm])
    assert abs(la.item() - lb.item()) <= 1e-5 * abs(la.item())
    l1 = MixedLoss(1.0, 0.0, 0.0)(p, t)
    assert abs(l1.item() - (p - t).abs().mean().item()) <= 1e-5 * l1.item()


# ------------------------------------------------------------------------------------------------ FCMAE dense path (§8 f2)
def test_baseline_size_fp32_and_bf16_engines_vs_reference_golden():
    """the production bf16 kernels (matrix-core depthwise, fused GRN-MLP passes, direct head convolution,
    both GEMM generations) pinned at the BASELINE patch size.  tests/golden/unext2_tiny_256.pt holds what the REFERENCE's own
    wiring (G8b, oracle/validate_against_reference.py) computes in fp32 for tiny, B = 4, 256 x 256: forward, MixedLoss value and
    a strided sample of every parameter gradient.  fp32 engine: forward <= 1e-3 of the output maximum, loss <= 1e-3, per-stage
    gradient direction 1 - cos < 2e-6.  bf16 engine: forward and every stage within 1.25 x the error of the oracle module under
    ``torch.autocast(bfloat16)`` — the reference's bf16-mixed arithmetic — computed here, on the GPU, against the same fp32
    golden (the CPU autocast backward is not reproducible run to run, so it is not a fixture)."""
    from oracle import loss_ref, unext2_ref
    from viscy_amd.losses import MixedLoss
    from viscy_amd.unext2 import UNeXt2

    gold = load_golden("unext2_tiny_256.pt")
    kw = gold["kwargs"]
    B, S = gold["shape"]
    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=gold["seed"])
    g = torch.Generator().manual_seed(gold["x_seed"])
    x = torch.randn((B, 1, 5, S, S), generator=g)
    # the target of the fixture: the module's own fp32 output + tgt_noise standard deviations of noise (why: the generator,
    # g8b_baseline_size — a target independent of the prediction puts ms_ssim_25d's 1e-4 clamp AT the operating point and the
    # reference's own gradient becomes discontinuous in the bf16 noise of the prediction)
    with torch.no_grad():
        y0 = ref(x)
    tgt = (y0 + gold["tgt_noise"] * y0.std() * torch.randn((B, 2, 5, S, S), generator=g)).contiguous()
    st = gold["y_stride"]

    def score(y, loss, grad_of):
        fwd = ((y.detach().float().cpu()[..., ::st, ::st] - gold["y"]).abs().max() / gold["y_absmax"]).item()
        stages = {}
        for gname, names in gold["groups"].items():
            a, b = [], []
            for n in names:
                stride, sample = gold["grad_samples"][n]
                a.append(sample.double())
                b.append(grad_of(n).flatten()[::stride].double().cpu())
            a, b = torch.cat(a), torch.cat(b)
            stages[gname] = (1.0 - torch.nn.functional.cosine_similarity(a, b, dim=0).item(), ((a - b).norm() / a.norm()).item())
        return fwd, abs(float(loss.detach()) - gold["loss"]) / abs(gold["loss"]), stages

    def run_engine(dt):
        m = UNeXt2(**kw)
        m.load_state_dict(ref.state_dict(), strict=True)
        m = m.cuda()
        m.compute_dtype, m.grad_mode = dt, "flat"
        eng = m.engine()
        eng.flat_grad.zero_()
        y = m(x.cuda())
        loss = MixedLoss(0.5, 0.0, 0.5)(y, tgt.cuda())
        loss.backward()
        named = dict(m.named_parameters())
        return score(y, loss, lambda n: eng.g(named[n]))

    def run_autocast_yardstick():
        o = unext2_ref.UNeXt2(**kw)
        o.load_state_dict(ref.state_dict(), strict=True)
        o = o.cuda()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = o(x.cuda())
        # the loss and its gradient in fp32 on the host (oracle/loss_ref.py is CPU code), as bf16-mixed keeps the loss in fp32
        yc = y.detach().float().cpu().requires_grad_(True)
        loss = loss_ref.mixed_loss(yc, tgt, 0.5, 0.0, 0.5)
        loss.backward()
        y.backward(yc.grad.to(device=y.device, dtype=y.dtype))
        named = dict(o.named_parameters())
        return score(y, loss.detach(), lambda n: named[n].grad)

    fwd, lrel, stages = run_engine(torch.float32)
    print("fp32 engine @256: forward", f"{fwd:.2e}", "loss", f"{lrel:.2e}", {k: f"{v[0]:.1e}/{v[1]:.1e}" for k, v in stages.items()})
    assert fwd <= 1e-3 and lrel <= 1e-3
    for gname, (omc, rel) in stages.items():
        assert omc < 2e-6 and rel < 4e-3, (gname, omc, rel)  # measured: <= 3.3e-7 / 8.1e-4
    yf, yl, ys = run_autocast_yardstick()
    fwd, lrel, stages = run_engine(torch.bfloat16)
    print("bf16 engine @256: forward", f"{fwd:.2e}", "(autocast", f"{yf:.2e})", "loss", f"{lrel:.2e}", f"({yl:.2e})",
          {k: f"{v[0]:.1e} (ac {ys[k][0]:.1e})" for k, v in stages.items()})
    # With a target the loss is differentiable at (see the fixture's generator) both sides repeat to +-5 % run to run: engine
    # per-stage 1 - cos 1.5e-3 .. 5.7e-3 against 1.8e-3 .. 7.5e-3 for the reference arithmetic under autocast, engine below
    # the yardstick in every stage of every run so far (ratio 0.6 .. 0.9).  (With the first fixture's independent target the
    # SAME code gave 2e-4 .. 3e-1 on both sides: ms_ssim_25d's 1e-4 clamp sat at the operating point)
    assert fwd <= 1.25 * yf
    assert lrel <= max(1.25 * yl, 1e-3)
    for gname, (omc, rel) in stages.items():
        assert omc <= 1.25 * ys[gname][0], (gname, omc, ys[gname])
        assert rel <= 1.25 * ys[gname][1], (gname, rel, ys[gname])
    # the block schedules behind the other values of `mlp_fused` are held to the same yardstick: 239 = the normalised rows x^ are
    # NOT stored but re-formed from the depthwise output (MODE 7; the default of rounds 4 - 5; round 6 ships 111 = x^ stored,
    # MODE 6 / 5: 0.6 % faster on the step), 47 = the pre-activation h is stored as well (MODE 2 / 4), 15 = LayerNorm as a pass
    # of its own
    from viscy_amd import _lib as L

    saved = L.lib().vsx_get_flag(b"mlp_fused")
    assert saved == 111, "the shipped schedule stores the normalised rows and recomputes h"
    try:
        for flag in (239, 47, 15):
            L.lib().vsx_set_flag(b"mlp_fused", flag)
            fwd, lrel, stages = run_engine(torch.bfloat16)
            print(f"bf16 engine @256, mlp_fused = {flag}: forward {fwd:.2e} loss {lrel:.2e}", {k: f"{v[0]:.1e}" for k, v in stages.items()})
            assert fwd <= 1.25 * yf and lrel <= max(1.25 * yl, 1e-3)
            for gname, (omc, rel) in stages.items():
                assert omc <= 1.25 * ys[gname][0] and rel <= 1.25 * ys[gname][1], (flag, gname, omc, rel, ys[gname])
    finally:
        L.lib().vsx_set_flag(b"mlp_fused", saved)


def test_gate_shape_fp32_and_bf16_engines_vs_reference_golden():
    """the north star's gate shape pinned on the REFERENCE, not only on properties.  tests/golden/
    unext2_tiny_2048.pt (G8c, oracle/validate_against_reference.py) holds what the reference's own wiring computes in fp32 on the
    CPU for tiny, B = 1, 2048 x 2048: a strided sample of the output and the per-sample GRN statistics ||h||_2 of one encoder and
    one decoder block.  fp32 engine: <= 1e-3 on both; bf16 engine (the kernel selections of the large maps: grid caps, split
    counts, per-sample fc2 weights, sub-split weight gradients are not exercised at 256 x 256) <= 1.25 x the error of the oracle
    module under torch.autocast(bfloat16), computed here on the GPU against the same fixture."""
    from viscy_amd.unext2 import UNeXt2

    gold = load_golden("unext2_tiny_2048.pt")
    kw = gold["kwargs"]
    B, S = gold["shape"]
    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=gold["seed"]).eval()
    x = torch.randn((B, 1, 5, S, S), generator=torch.Generator().manual_seed(gold["x_seed"]))
    st = gold["y_stride"]

    def score(y, grn):
        fwd = ((y.detach().float().cpu()[..., ::st, ::st] - gold["y"]).abs().max() / gold["y_absmax"]).item()
        gs = {k: ((grn[k].float().cpu() - v).abs().max() / v.abs().max()).item() for k, v in gold["grn"].items()}
        return fwd, gs

    def run_engine(dt):
        m = UNeXt2(**kw)
        m.load_state_dict(ref.state_dict(), strict=True)
        m = m.cuda()
        m.compute_dtype, m.grad_mode = dt, "flat"
        eng = m.engine()
        y, sv = eng.forward(x.cuda(), dt, need_bwd=True)  # the training schedule: the blocks' GRN sums are in the saved state
        grn = {}
        for tag, (part, si, bi) in gold["grn_paths"].items():
            colsq = sv[part][si]["blocks"][bi][5]   # [B, 4C] sum over the sample's pixels of gelu(h)^2
            grn[tag] = colsq.sqrt()
        out = score(y, grn)
        eng._pending_bwd = 0
        del sv, y, m, eng
        torch.cuda.empty_cache()
        return out

    def run_autocast_yardstick():
        o = unext2_ref.UNeXt2(**kw)
        o.load_state_dict(ref.state_dict(), strict=True)
        o = o.cuda().eval()
        grn, hooks = {}, []
        mods = {"enc_s0_b1": o.encoder_stages.stages_0.blocks[1].mlp.grn, "dec_s2_b0": o.decoder.decoder_stages[2].conv.blocks[0].mlp.grn}
        for tag, mod in mods.items():
            hooks.append(mod.register_forward_hook(
                lambda md, inp, out, tag=tag: grn.__setitem__(tag, inp[0].float().norm(p=2, dim=md.spatial_dim).reshape(1, -1))))
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            y = o(x.cuda())
        for h in hooks:
            h.remove()
        out = score(y, grn)
        del o, y
        torch.cuda.empty_cache()
        return out

    fwd, gs = run_engine(torch.float32)
    print("fp32 engine @2048: forward", f"{fwd:.2e}", "GRN", {k: f"{v:.1e}" for k, v in gs.items()})
    assert fwd <= 1e-3, fwd
    for k, v in gs.items():
        assert v <= 1e-3, (k, v)
    yf, yg = run_autocast_yardstick()
    fwd, gs = run_engine(torch.bfloat16)
    print("bf16 engine @2048: forward", f"{fwd:.2e}", "(autocast", f"{yf:.2e})", "GRN", {k: f"{v:.1e} (ac {yg[k]:.1e})" for k, v in gs.items()})
    assert fwd <= 1.25 * yf, (fwd, yf)
    for k, v in gs.items():
        # the statistic is a sum over 262 144 pixels: under autocast its error is the bf16 rounding of the operands averaged out
        # (a few 1e-4); the engine's must stay in that class
        assert v <= max(1.25 * yg[k], 2e-3), (k, v, yg[k])


def test_gate_shape_gradients_vs_reference_golden():
    """the BACKWARD at the north star's gate shape pinned on the reference.  G8c (tests/golden/
    unext2_tiny_2048.pt) holds, for tiny at B = 1, 2048 x 2048, strided samples of the reference's own fp32 gradient of the linear
    functional <y, c> (c: seeded) with respect to five parameters — at least one in each gradient bucket of the engine.  fp32
    engine: every sample within 1e-3 of the tensor's largest gradient; bf16 engine (the production kernels at the large-map
    selections: grid caps, sub-split weight gradients, per-sample products with sub-splits + atomics): no further from the
    reference than 1.25 x the oracle module under torch.autocast(bfloat16) on the same input (direction: 1 - cos of the sample;
    size: relative error of the sample)."""
    from viscy_amd.unext2 import UNeXt2

    gold = load_golden("unext2_tiny_2048.pt")
    if "grads" not in gold:
        pytest.skip("fixture without gradient samples (regenerate with oracle/validate_against_reference.py)")
    kw = gold["kwargs"]
    B, S = gold["shape"]
    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=gold["seed"]).eval()
    x = torch.randn((B, 1, 5, S, S), generator=torch.Generator().manual_seed(gold["x_seed"])).cuda()
    ysh = (B, kw["out_channels"], 5, S, S)
    n = 1
    for d in ysh:
        n *= d
    cot = (torch.randn(ysh, generator=torch.Generator().manual_seed(gold["cot_seed"])) / n ** 0.5).cuda()

    def sample_of(named_grads):
        out = {}
        for name, gs in gold["grads"].items():
            out[name] = named_grads[name].detach().float().flatten()[:: gs["stride"]].cpu()
        return out

    def errors(samples):
        e = {}
        for name, gs in gold["grads"].items():
            a, b_ = samples[name].double(), gs["sample"].double()
            e[name] = ((a - b_).abs().max().item() / gs["absmax"], 1.0 - torch.nn.functional.cosine_similarity(a, b_, dim=0).item(),
                       ((a - b_).norm() / b_.norm()).item())
        return e

    def run_engine(dt):
        m = UNeXt2(**kw)
        m.load_state_dict(ref.state_dict(), strict=True)
        m = m.cuda()
        m.compute_dtype, m.grad_mode = dt, "flat"
        eng = m.engine()
        eng.flat_grad.zero_()
        y = m(x)
        (y * cot).sum().backward()
        out = errors(sample_of({k: eng.g(p) for k, p in m.named_parameters()}))
        del y, m, eng
        torch.cuda.empty_cache()
        return out

    def run_autocast_yardstick():
        o = unext2_ref.UNeXt2(**kw)
        o.load_state_dict(ref.state_dict(), strict=True)
        o = o.cuda().eval()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = o(x)
        (y.float() * cot).sum().backward()
        out = errors(sample_of({k: p.grad for k, p in o.named_parameters()}))
        del o, y
        torch.cuda.empty_cache()
        return out

    e32 = run_engine(torch.float32)
    for name, (mx, oc, rel) in e32.items():
        print(f"fp32 engine @2048 d/d {name}: max err / max |g| {mx:.2e}  1-cos {oc:.2e}  rel {rel:.2e}")
    yard = run_autocast_yardstick()
    e16 = run_engine(torch.bfloat16)
    for name, (mx, oc, rel) in e16.items():
        ymx, yoc, yrel = yard[name]
        print(f"bf16 engine @2048 d/d {name}: 1-cos {oc:.2e} (autocast {yoc:.2e})  rel {rel:.2e} (autocast {yrel:.2e})")
    # The fp32 bar.  The forward holds 1e-3 at this size (test above: 4e-6).  A GRADIENT of this network in fp32 does not, on either
    # side: InstanceNorm outputs within rounding of 0 fall on either side of the PReLU kink, and every flipped voxel moves the whole
    # upstream gradient.  The reference's own fp32 gradient deviates from its fp64 gradient by 1.2e-4 at 256^2, 5e-4 at 512^2 and
    # ~1e-3 at 1024^2 (G8d, tests/golden/unext2_tiny_1024_fp64.pt; the test below holds the engine to THAT yardstick at 1024^2,
    # where an fp64 truth is affordable).  Here, against the fp32 golden (itself ~2e-3 from the truth): direction to 2e-6 and
    # size to 3e-3 — two fp32 evaluations of the same gradient, each within its own noise of the truth.
    for name, (mx, oc, rel) in e32.items():
        assert oc <= 2e-6 and rel <= 3e-3 and mx <= 4e-3, (name, mx, oc, rel)
    for name, (mx, oc, rel) in e16.items():
        ymx, yoc, yrel = yard[name]
        assert oc <= 1.25 * yoc + 1e-6 and rel <= 1.25 * yrel + 1e-4, (name, oc, yoc, rel, yrel)


def test_det_reduce_makes_the_bf16_forward_bit_identical_from_run_to_run():
    """with fp32 atomics the GRN / InstanceNorm sums are added in an order that differs from run to run,
    and a bf16 forward at a large image differs by ~1e-3 .. 1e-2 of its maximum between two runs of the same weights and input
    (DESIGN §5).  `vsx_set_flag("det_reduce", 1)` forms those sums in a fixed order (per-workgroup partials + one ordered pass):
    three forwards must agree BIT FOR BIT, and with the reduction-order noise of the atomic path to the tolerance that noise has."""
    from viscy_amd import _lib as L
    from viscy_amd.unext2 import UNeXt2

    kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4,
              decoder_conv_blocks=2)
    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=9)
    m = UNeXt2(**kw)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda()
    m.compute_dtype, m.grad_mode = torch.bfloat16, "flat"
    eng = m.engine()
    x = torch.randn((2, 1, 5, 1024, 1024), generator=torch.Generator().manual_seed(3)).cuda()   # 16 - 64 workgroups per sample and stage

    def fwd(training):
        y, sv = eng.forward(x, torch.bfloat16, need_bwd=training)
        eng._pending_bwd = 0
        return y.clone()

    lib = L.lib()
    try:
        lib.vsx_set_flag(b"det_reduce", 1)
        for training in (True, False):          # training schedule (MODE 6 / 2) and inference schedule (MODE 0 / 1)
            y0 = fwd(training)
            for _ in range(2):
                assert torch.equal(fwd(training), y0), ("det_reduce forward differs between runs", training)
        y_det = fwd(True)
        lib.vsx_set_flag(b"det_reduce", 0)
        y_atomic = fwd(True)
    finally:
        lib.vsx_set_flag(b"det_reduce", 0)
    d = ((y_det.float() - y_atomic.float()).abs().max() / y_det.float().abs().max()).item()
    assert d <= 2e-2, d   # the same sums in another order: bf16 rounding noise, not a different result
    # round 6: a forward with no backward behind it (predict / validation) takes the fixed order by itself, flag untouched
    assert lib.vsx_get_flag(b"det_reduce") == 0
    y0 = fwd(False)
    assert torch.equal(fwd(False), y0) and torch.equal(fwd(False), y0)
    assert lib.vsx_get_flag(b"det_reduce") == 0


def test_large_image_fp32_gradient_is_as_accurate_as_the_reference_fp32_arithmetic():
    """G8d: at 1024 x 1024 (tiny, B = 1) the fixture holds strided samples of the reference's fp64 gradient of <y, c> and the
    deviation of the reference's OWN fp32 gradient from it (up to ~1e-3: PReLU-kink flips, see the generator's docstring).  The
    fp32 engine must be no further from the fp64 gradient than 1.5 x that (and never further than 3e-3): the parity bar for a
    quantity that the reference itself only knows to 1e-3."""
    from viscy_amd.unext2 import UNeXt2

    gold = load_golden("unext2_tiny_1024_fp64.pt")
    kw = gold["kwargs"]
    B, S = gold["shape"]
    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=gold["seed"]).eval()
    x = torch.randn((B, 1, 5, S, S), generator=torch.Generator().manual_seed(gold["x_seed"])).cuda()
    ysh = (B, kw["out_channels"], 5, S, S)
    n = 1
    for d in ysh:
        n *= d
    cot = (torch.randn(ysh, generator=torch.Generator().manual_seed(gold["cot_seed"])) / n ** 0.5).cuda()
    m = UNeXt2(**kw)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda()
    m.compute_dtype, m.grad_mode = torch.float32, "flat"
    eng = m.engine()
    eng.flat_grad.zero_()
    y = m(x)
    (y * cot).sum().backward()
    named = dict(m.named_parameters())
    for name, gs in gold["grads"].items():
        a = eng.g(named[name]).detach().double().flatten()[:: gs["stride"]].cpu()
        b_ = gs["sample64"]
        rel = ((a - b_).norm() / b_.norm()).item()
        oc = 1.0 - torch.nn.functional.cosine_similarity(a, b_, dim=0).item()
        print(f"fp32 engine @1024 d/d {name}: vs fp64 reference rel {rel:.2e} (reference fp32: {gs['ref32_rel']:.2e})  1-cos {oc:.2e} "
              f"(reference fp32: {gs['ref32_1mcos']:.2e})")
        assert rel <= max(1.5 * gs["ref32_rel"], 2e-4) and rel <= 3e-3, (name, rel, gs["ref32_rel"])
        assert oc <= max(2.25 * gs["ref32_1mcos"], 4e-8), (name, oc, gs["ref32_1mcos"])


def test_large_batch_gradient_equals_sum_of_pinned_small_batches():
    """VERDICT r3 item 7, second half: the bf16 engine is pinned against the reference at B = 4 (above); the bench runs B = 512, where
    other kernel selections are taken (tile counts decide between GEMM instantiations, split counts, grid caps, the per-sample
    weight products).  One bf16 forward / backward at B = 64, 256 x 256 (loss = <y, fixed cotangent>, linear in y, so that
    per-sample gradients add) against the SUM of the gradients of 16 passes at B = 4 — the configuration the golden pins.  The
    two are the same sum with other fp32 accumulation orders in front of the bf16 roundings, i.e. two draws of the bf16 rounding
    noise: they are compared through the fp32 engine's gradient of the same batch (the engine the golden holds to 1e-3 / 2e-6):
    the large batch may be no further from it than the pinned small batches are (x 1.25), bucket by bucket."""
    from viscy_amd.unext2 import UNeXt2

    kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4,
              decoder_conv_blocks=2)
    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=21)
    g = torch.Generator().manual_seed(5)
    Bb, S = 64, 256
    x = torch.randn((Bb, 1, 5, S, S), generator=g).cuda()
    cot = torch.randn((Bb, 2, 5, S, S), generator=g).cuda() / (Bb * S)   # fixed cotangent: loss = <y, cot>

    def engine(dt):
        m = UNeXt2(**kw)
        m.load_state_dict(ref.state_dict(), strict=True)
        m = m.cuda()
        m.compute_dtype, m.grad_mode = dt, "flat"
        return m, m.engine()

    def grad_of(m, eng, lo, hi):
        eng.flat_grad.zero_()
        y = m(x[lo:hi])
        (y * cot[lo:hi]).sum().backward()
        return eng.flat_grad.double().clone()

    m32, e32 = engine(torch.float32)
    g32 = grad_of(m32, e32, 0, Bb)
    bounds = list(e32.bucket_bounds)
    del m32, e32
    torch.cuda.empty_cache()
    m, eng = engine(torch.bfloat16)
    big = grad_of(m, eng, 0, Bb)
    small = torch.zeros_like(big)
    for i in range(0, Bb, 4):
        small += grad_of(m, eng, i, i + 4)

    def dist(a, b):
        return 1.0 - torch.nn.functional.cosine_similarity(a, b, dim=0).item(), ((a - b).norm() / b.norm()).item()

    for (lo, hi) in bounds:  # head + decoder | encoder 3-2 | encoder 1-0 + stem
        ob, rb = dist(big[lo:hi], g32[lo:hi])
        os_, rs = dist(small[lo:hi], g32[lo:hi])
        oc, rc = dist(big[lo:hi], small[lo:hi])
        print(f"flat[{lo}:{hi}] vs fp32 engine: B=64 1-cos {ob:.2e} rel {rb:.2e} | 16 x B=4 1-cos {os_:.2e} rel {rs:.2e} | B=64 vs 16 x B=4 {oc:.2e} / {rc:.2e}")
        assert ob <= 1.25 * os_ and rb <= 1.25 * rs, (lo, hi, ob, os_, rb, rs)
        assert oc <= 2.0 * os_, (lo, hi, oc, os_)   # two draws of the same noise: not further apart than ~2 x one draw from fp32


@pytest.mark.parametrize("tag", ["small_z5", "vscyto3d_z15", "head_conv_z5"])
def test_fcmae_forward_matches_reference_golden_fp32(tag):
    """fixtures produced by the REFERENCE's own fcmae.py (oracle/validate_against_reference.py G9)"""
    from oracle import fcmae_ref
    from viscy_amd.fcmae import FullyConvolutionalMAE

    g = load_golden("fcmae_forward.pt")[tag]
    ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**g["kwargs"]), seed=g["seed"]).eval()
    mine = FullyConvolutionalMAE(**g["kwargs"])
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    mine.compute_dtype = torch.float32
    x = torch.randn(g["x_shape"], generator=torch.Generator().manual_seed(g["x_seed"]))
    with torch.no_grad():
        y = mine(x.cuda())
        y2d = mine(x[:, :, :1].contiguous().cuda())  # Z == 1 input: the Conv2d stem branch (fcmae.py:369-370)
    assert y.shape == g["y"].shape and y.dtype == torch.float32
    assert relerr(y, g["y"]) <= 1e-3
    assert y2d.shape == g["y_2d"].shape and relerr(y2d, g["y_2d"]) <= 1e-3


@pytest.mark.parametrize("flat_input", [False, True], ids=["zstack", "z1"])
@pytest.mark.parametrize("tag", ["small_z5", "head_conv_z5"])
def test_fcmae_forward_backward_vs_oracle_fp32(tag, flat_input):
    from oracle import fcmae_ref
    from viscy_amd.fcmae import FullyConvolutionalMAE

    g = load_golden("fcmae_forward.pt")[tag]
    kw = g["kwargs"]
    ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**kw), seed=3).eval()
    if tag == "head_conv_z5":
        with torch.no_grad():
            ref.head.conv[0].adn.A.weight.fill_(1.0)  # no PReLU kink (see test_forward_backward_vs_oracle_fp32)
    mine = FullyConvolutionalMAE(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    mine.compute_dtype = torch.float32
    x = torch.randn(2, kw["in_channels"], 1 if flat_input else kw["in_stack_depth"], 64, 96,
                    generator=torch.Generator().manual_seed(5))
    y = ref(x)
    dout = torch.randn(y.shape, generator=torch.Generator().manual_seed(6))
    y.backward(dout)
    out = mine(x.cuda())
    assert relerr(out, y) <= 1e-3
    out.backward(dout.cuda())
    worst = 0.0
    for (name, pr), (n2, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        assert name == n2
        if pr.grad is None:  # the stem branch that is not on this input's path (Conv2d for a Z-stack, Conv3d for Z == 1)
            assert ("conv3d" if flat_input else "conv2d") in name
            assert pm.grad is None or float(pm.grad.abs().max()) == 0.0
            continue
        if name == "head.conv.0.conv.bias":
            continue
        e = relerr(pm.grad, pr.grad)
        worst = max(worst, e)
        assert e <= 2e-3, (name, e)
    print(tag, "fcmae worst relative gradient error", worst)


def test_fcmae_vscyto3d_bf16_tracks_fp32_and_trains():
    """the published VSCyto3D configuration (Z = 15, shuffle head) in production precision: forward within the reference's
    GPU reproducibility tolerance of the fp32 oracle, and a few fused AdamW steps through VSUNet(architecture="fcmae")
    reduce the loss."""
    from oracle import fcmae_ref
    from viscy_amd.losses import MixedLoss
    from viscy_amd.vsunet import VSUNet

    kw = dict(in_channels=1, out_channels=2, encoder_blocks=[3, 3, 9, 3], dims=[96, 192, 384, 768], decoder_conv_blocks=2,
              stem_kernel_size=(5, 4, 4), in_stack_depth=15, pretraining=False)
    ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**kw), seed=9).eval()
    vs = VSUNet("fcmae", kw, loss_function=MixedLoss(0.5, 0.0, 0.5), lr=2e-4)
    vs.model.load_state_dict(ref.state_dict(), strict=True)
    vs = vs.cuda()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 1, 15, 192, 192, generator=g)
    with torch.no_grad():
        y = ref(x)
        vs.model.compute_dtype = torch.bfloat16
        out = vs(x.cuda())
    torch.testing.assert_close(out.cpu(), y, rtol=1e-2, atol=0.02 * y.abs().max().item())
    tgt = torch.rand(2, 2, 15, 192, 192, generator=g).cuda()
    opt = vs.configure_optimizers(t_total=10)
    losses = []
    for i in range(10):
        opt.zero_grad()
        loss = vs.training_step({"source": x.cuda(), "target": tgt}, i)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(l == l for l in losses) and min(losses[5:]) < losses[0], losses


# ------------------------------------------------------------------------------------------------ FCMAE masked pre-training (§8 f2)
def _masked_pair(tag, dtype):
    from oracle import fcmae_ref
    from viscy_amd.fcmae import FullyConvolutionalMAE

    g = load_golden("fcmae_masked.pt")[tag]
    kw = g["kwargs"]
    ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**kw), seed=g["seed"])
    mine = FullyConvolutionalMAE(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    mine.compute_dtype = dtype
    x = torch.randn(g["x_shape"], generator=torch.Generator().manual_seed(g["x_seed"]))
    return g, ref, mine, x


@pytest.mark.parametrize("tag", ["small_z5_r50", "two_ch_r75"])
def test_fcmae_masked_matches_reference_golden_fp32(tag):
    """masked forward, MaskedMSELoss value and parameter gradients vs the REFERENCE's own run (generate_mask draw, fcmae.py
    masked path, cytoland MaskedMSELoss; oracle/validate_against_reference.py G9) — the same mask is injected."""
    from viscy_amd.losses import MaskedMSELoss

    g, ref, mine, x = _masked_pair(tag, torch.float32)
    xc = x.cuda()
    y, mask = mine(xc, mask=g["mask_low"].cuda())
    assert mask.dtype == torch.bool and tuple(mask.shape) == (x.shape[0], 1, x.shape[-2], x.shape[-1])
    assert abs(mask.float().mean().item() - g["mask_ratio"]) < 1e-6
    assert relerr(y, g["y"]) <= 1e-3
    loss = MaskedMSELoss()(y, xc, mask)
    assert abs(loss.item() - g["loss"]) <= 1e-4 * abs(g["loss"])
    loss.backward()
    named = dict(mine.named_parameters())
    for name, gg in g["grads"].items():
        assert relerr(named[name].grad, gg) <= 2e-3, name


def test_fcmae_masked_all_gradients_vs_oracle_fp32():
    from oracle import fcmae_ref
    from viscy_amd.losses import MaskedMSELoss

    g, ref, mine, x = _masked_pair("two_ch_r75", torch.float32)
    low = g["mask_low"]
    y, mask = ref(x, mask=low)
    fcmae_ref.MaskedMSELoss()(y, x, mask).backward()
    out, m2 = mine(x.cuda(), mask=low.cuda())
    assert torch.equal(m2.cpu(), mask)
    MaskedMSELoss()(out, x.cuda(), m2).backward()
    worst = 0.0
    for (name, pr), (n2, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        assert name == n2
        if pr.grad is None:
            continue
        e = relerr(pm.grad, pr.grad)
        worst = max(worst, e)
        assert e <= 2e-3, (name, e)
    print("fcmae masked worst relative gradient error", worst)


def test_fcmae_generated_mask_properties_and_bf16():
    """mask_ratio path: exactly int(n * ratio) cells hidden per sample at stride 32; the kept tokens of a bf16 run track the
    fp32 oracle fed the same mask; masked encoder inputs do not influence the prediction (the encoder never sees them)."""
    from oracle import fcmae_ref

    g, ref, mine, x = _masked_pair("small_z5_r50", torch.bfloat16)
    xc = x.cuda()
    torch.manual_seed(3)
    with torch.no_grad():
        y, mask = mine(xc, mask_ratio=0.6)
    B, H, W = x.shape[0], x.shape[-2], x.shape[-1]
    n = (H // 32) * (W // 32)
    cells = mask[:, 0, ::32, ::32]
    assert torch.equal(cells.repeat_interleave(32, 1).repeat_interleave(32, 2), mask[:, 0])
    assert cells.flatten(1).sum(1).tolist() == [int(n * 0.6)] * B
    with torch.no_grad():
        yr, mr = ref(x, mask=cells.unsqueeze(1).cpu())
    assert torch.equal(mr, mask.cpu())
    torch.testing.assert_close(y.cpu(), yr, rtol=2e-2, atol=0.03 * yr.abs().max().item())
    # hidden voxels are invisible to the network
    x2 = torch.where(mask.unsqueeze(2), torch.randn_like(xc), xc)
    with torch.no_grad():
        y2, _ = mine(x2, mask=cells.unsqueeze(1))
    torch.testing.assert_close(y2, y, rtol=0, atol=0.01 * y.abs().max().item())  # run-to-run bf16 noise only


def test_fcmae_unet_pretraining_steps_reduce_the_masked_loss():
    """cytoland FcmaeUNet pre-training recipe: fit_mask_ratio 0.5, MaskedMSELoss, fused AdamW, bf16."""
    from viscy_amd.losses import MaskedMSELoss, MixedLoss
    from viscy_amd.vsunet import FcmaeUNet

    kw = dict(in_channels=1, out_channels=1, encoder_blocks=[2, 2, 2, 2], dims=[32, 64, 128, 256], decoder_conv_blocks=1,
              in_stack_depth=5, pretraining=True)
    with pytest.raises(ValueError, match="requires ckpt_path"):
        FcmaeUNet(encoder_only=True, model_config=kw)
    bad = FcmaeUNet(fit_mask_ratio=0.5, model_config=kw, loss_function=MixedLoss())
    with pytest.raises(ValueError, match="MaskedMSELoss is required"):
        bad.on_fit_start()
    vs = FcmaeUNet(fit_mask_ratio=0.5, model_config=kw, loss_function=MaskedMSELoss(), lr=1e-3).cuda()
    vs.on_fit_start()
    vs.model.compute_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(4)
    base = torch.nn.functional.avg_pool3d(torch.randn(4, 1, 5, 128, 128, generator=g), (1, 9, 9), 1, (0, 4, 4)).cuda() * 4
    opt = vs.configure_optimizers(t_total=30)
    losses = []
    for i in range(30):
        opt.zero_grad()
        loss = vs.training_step([{"source": base[:2]}, {"source": base[2:]}], i)  # a CombinedLoader-style list of batches
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(l == l for l in losses)
    assert sum(losses[-5:]) / 5 < 0.8 * sum(losses[:5]) / 5, losses
    with torch.no_grad():
        vs.validation_step({"source": base}, 0)
    assert len(vs.validation_losses[0]) == 1


# ------------------------------------------------------------------------------------------------ DynaCLR path (§8 f3)
def _contrastive_pair(dtype, tag="v2_small_z9"):
    from oracle import contrastive_ref as C
    from viscy_amd.contrastive import ContrastiveEncoder

    g = load_golden("contrastive.pt")[tag]
    ref = C.randomize_encoder_(C.ContrastiveEncoder(**g["kwargs"], **g["arch"]), seed=g["seed"])
    mine = ContrastiveEncoder(**g["kwargs"], **g["arch"])
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    mine.compute_dtype = dtype
    x = torch.randn(g["x_shape"], generator=torch.Generator().manual_seed(g["x_seed"]))
    return g, ref, mine, x


@pytest.mark.parametrize("tag", ["v2_small_z9", "v1_small_z5", "v1_tiny_z15"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_contrastive_encoder_matches_reference_golden_fp32(mode, tag):
    """(embedding, projection) vs the REFERENCE's encoder.py run (G10; V2 = GRN blocks, V1 = layer-scale blocks, the full
    convnext_tiny trunk at the DynaCLR shape 2 ch x 15 slices), BatchNorm running statistics after a train-mode call, every
    parameter gradient vs oracle autograd"""
    g, ref, mine, x = _contrastive_pair(torch.float32, tag)
    getattr(ref, mode)()
    getattr(mine, mode)()
    emb, proj = mine(x.cuda())
    assert relerr(emb, g[mode][0]) <= 1e-3 and relerr(proj, g[mode][1]) <= 1e-3
    if mode == "train":
        sd = mine.state_dict()
        for k, v in g["running_after"].items():
            torch.testing.assert_close(sd[k].cpu().float(), v.float(), rtol=1e-3, atol=1e-4)
    er, pr = ref(x)
    gen = torch.Generator().manual_seed(3)
    de, dp = torch.randn(er.shape, generator=gen), torch.randn(pr.shape, generator=gen)
    ((er * de).sum() + (pr * dp).sum()).backward()
    ((emb * de.cuda()).sum() + (proj * dp.cuda()).sum()).backward()
    worst = 0.0
    for (name, p_ref), (n2, p) in zip(ref.named_parameters(), mine.named_parameters()):
        assert name == n2
        if mode == "train" and name in ("projection.0.bias", "projection.3.bias"):
            continue  # exactly-zero gradient in front of a train-mode BatchNorm
        e = relerr(p.grad, p_ref.grad)
        worst = max(worst, e)
        assert e <= 2e-3, (name, e)
    print(tag, mode, "contrastive worst relative gradient error", worst)


def test_contrastive_module_trains_convnextv2_tiny_bf16():
    """the DynaCLR recipe shape: convnextv2_tiny trunk, 2 channels x 15 slices, NT-Xent on (anchor, positive) pairs, fused
    AdamW, bf16 — the loss falls and the embeddings of a pair become the nearest neighbours of each other"""
    from viscy_amd.contrastive import ContrastiveEncoder, ContrastiveModule, NTXentLoss

    torch.manual_seed(0)
    enc = ContrastiveEncoder("convnextv2_tiny", in_channels=2, in_stack_depth=15, embedding_dim=768, projection_dim=128)
    mod = ContrastiveModule(enc, loss_function=NTXentLoss(temperature=0.2), lr=2e-4).cuda()
    enc.compute_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    base = torch.nn.functional.avg_pool3d(torch.randn(8, 2, 15, 96, 96, generator=g), (1, 5, 5), 1, (0, 2, 2)).cuda() * 3
    batch = {"anchor": base, "positive": base + 0.5 * torch.randn(base.shape, generator=g).cuda()}
    with torch.no_grad():
        emb, proj = mod(base)
    assert emb.shape == (8, 768) and proj.shape == (8, 128) and emb.dtype == torch.float32
    opt = mod.configure_optimizers(t_total=20)
    mod.train()
    mod.on_train_epoch_start()
    losses = []
    for i in range(20):
        opt.zero_grad()
        loss = mod.training_step(batch, i)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(l == l for l in losses) and sum(losses[-3:]) / 3 < 0.9 * losses[0], losses
    mod.eval()
    with torch.no_grad():
        pa = torch.nn.functional.normalize(mod.predict_step({"anchor": batch["anchor"]}, 0)["projections"], dim=1)
        pp = torch.nn.functional.normalize(mod.predict_step({"anchor": batch["positive"]}, 0)["projections"], dim=1)
    assert (pa @ pp.t()).argmax(1).tolist() == list(range(8))
    with pytest.raises(NotImplementedError, match="resnet50"):
        ContrastiveEncoder("resnet50", in_channels=2, in_stack_depth=15)


def test_graph_captured_contrastive_and_pretraining_steps_match_eager():
    """TrainStep(loss_fn=...): the DynaCLR step and the FCMAE masked pre-training step replayed as one hipGraph give the same
    trajectory as eager launches (same seeds; the mask draw is device-side randomness inside the capture)"""
    from viscy_amd.contrastive import ContrastiveEncoder, ContrastiveModule, NTXentLoss
    from viscy_amd.losses import MaskedMSELoss
    from viscy_amd.vsunet import FcmaeUNet

    g = torch.Generator().manual_seed(1)
    a = torch.randn(4, 1, 5, 64, 64, generator=g).cuda()
    p = a + 0.3 * torch.randn(a.shape, generator=g).cuda()
    traj = []
    for use_graph in (False, True):
        torch.manual_seed(0)
        enc = ContrastiveEncoder("convnext_tiny", in_channels=1, in_stack_depth=5, embedding_dim=64, projection_dim=32,
                                 depths=(1, 1, 2, 1), dims=(32, 64, 96, 128))
        mod = ContrastiveModule(enc, loss_function=NTXentLoss(temperature=0.3), lr=1e-3).cuda()
        enc.compute_dtype = torch.float32
        opt = mod.configure_optimizers(t_total=8)
        mod.train()
        step = mod.make_train_step(opt, use_graph=use_graph)
        traj.append([step(a, p).item() for _ in range(8 if not use_graph else 6)])
        # two forwards per step; the capture's warm-up steps are undone (parameters, moments, step counter AND the
        # BatchNorm buffers are restored): N calls are N optimisation steps in both modes
        assert int(enc.projection[1].num_batches_tracked) == (16 if not use_graph else 12)
    assert all(abs(x - y) <= 2e-3 * abs(x) + 1e-5 for x, y in zip(traj[0][:6], traj[1])), traj
    assert traj[0][-1] < traj[0][0]
    # FCMAE masked pre-training under capture: finite, decreasing, and a fresh mask every replay
    kw = dict(in_channels=1, out_channels=1, encoder_blocks=[1, 1, 1, 1], dims=[16, 32, 64, 128], decoder_conv_blocks=1,
              in_stack_depth=5, pretraining=True)
    vs = FcmaeUNet(fit_mask_ratio=0.5, model_config=kw, loss_function=MaskedMSELoss(), lr=1e-3).cuda()
    vs.model.compute_dtype = torch.bfloat16
    opt = vs.configure_optimizers(t_total=30)
    x = torch.nn.functional.avg_pool3d(torch.randn(4, 1, 5, 128, 128, generator=g), (1, 9, 9), 1, (0, 4, 4)).cuda() * 4
    step = vs.make_pretrain_step(opt)
    losses = [step(x, x).item() for _ in range(30)]
    assert all(l == l for l in losses) and len(set(losses)) > 20
    assert sum(losses[-5:]) < 0.8 * sum(losses[:5]), losses


def test_contrastive_paired_forward_matches_two_forwards():
    """ContrastiveModule: one trunk pass over [anchor; positive] (per-group BatchNorm) == the reference's two forwards — loss,
    gradients and BatchNorm running statistics"""
    from viscy_amd.contrastive import ContrastiveEncoder, ContrastiveModule, NTXentHCL

    g = torch.Generator().manual_seed(2)
    a = torch.randn(6, 2, 15, 64, 64, generator=g).cuda()
    batch = {"anchor": a, "positive": a + 0.3 * torch.randn(a.shape, generator=g).cuda()}
    res = []
    for paired in (True, False):
        torch.manual_seed(0)
        enc = ContrastiveEncoder("convnextv2_tiny", in_channels=2, in_stack_depth=15, embedding_dim=64, projection_dim=32,
                                 depths=(1, 1, 2, 1), dims=(24, 48, 96, 192))
        mod = ContrastiveModule(enc, loss_function=NTXentHCL(temperature=0.2, beta=0.3)).cuda().train()
        mod.paired_forward = paired
        enc.compute_dtype = torch.float32
        loss = mod.training_step(batch, 0)
        loss.backward()
        res.append((loss.item(), {n: p.grad.clone() for n, p in enc.named_parameters()},
                    {k: v.clone() for k, v in enc.state_dict().items() if "running" in k or "num_batches" in k}))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    for n in res[0][1]:
        if res[1][1][n].abs().max() < 1e-6:
            continue
        assert relerr(res[0][1][n], res[1][1][n]) <= 1e-3, n
    for k in res[0][2]:
        torch.testing.assert_close(res[0][2][k].float(), res[1][2][k].float(), rtol=1e-5, atol=1e-6)


def test_v1_backbone_fc2_bias_gradient_on_the_statistics_in_tn_path():
    """ConvNeXt-V1 blocks on the C = 384 stage with a per-rank batch whose per-sample products would not fit
    (B >= 228, 8 x 8 maps) take the fc2 weight gradient that also delivers the GRN statistics (gemm_tn_fast_kernel PRO == 2); the
    column sums of that launch must reach fc2.bias (and, through it, gamma) before `layer_scale_unfold` reads them.  Checked
    against the same backward with that launch switched off (`tn_rect` bit 3: statistics from the MODE 3 pass, colsum = db2)."""
    from viscy_amd import _lib as L
    from viscy_amd.contrastive import ContrastiveEncoder

    lib = L.lib()
    old = lib.vsx_get_flag(b"tn_rect")
    g = torch.Generator().manual_seed(3)
    B = 232
    a = torch.randn(B, 1, 5, 128, 128, generator=g).cuda()
    res = []
    try:
        for on in (True, False):
            lib.vsx_set_flag(b"tn_rect", (old | 8) if on else (old & ~8))
            torch.manual_seed(0)
            enc = ContrastiveEncoder("convnext_tiny", in_channels=1, in_stack_depth=5, embedding_dim=768, projection_dim=32,
                                     depths=(1, 1, 2, 1)).cuda().train()
            enc.compute_dtype = torch.bfloat16
            with torch.no_grad():  # a layer scale of 1e-6 would leave nothing to see in bf16
                for n, p in enc.named_parameters():
                    if n.endswith("gamma"):
                        p.fill_(0.5)
            emb, proj = enc(a)
            cot = torch.randn(proj.shape, generator=torch.Generator().manual_seed(4)).cuda()
            (proj * cot).sum().backward()
            res.append({n: p.grad.float().clone() for n, p in enc.named_parameters() if ".stages.2." in n or n.startswith("stages.2.")})
    finally:
        lib.vsx_set_flag(b"tn_rect", old)
    names = [n for n in res[0] if n.endswith("mlp.fc2.bias") or n.endswith("gamma")]
    assert len(names) == 4, list(res[0])
    for n in names:
        a_, b_ = res[0][n].flatten(), res[1][n].flatten()
        assert b_.abs().max() > 0 and a_.abs().max() > 0.2 * b_.abs().max(), n
        cos = torch.dot(a_, b_) / (a_.norm() * b_.norm())
        assert cos > 0.99, (n, float(cos))


def test_fcmae_finetune_with_frozen_encoder():
    """FcmaeUNet(freeze_encoder=True) (cytoland engine.py:204-206 + the FCMAE fine-tuning recipe): the backward stops after
    the decoder, the fused AdamW leaves encoder + stem untouched, decoder gradients equal those of the unfrozen run"""
    from viscy_amd.losses import MixedLoss
    from viscy_amd.vsunet import FcmaeUNet

    kw = dict(in_channels=1, out_channels=2, encoder_blocks=[1, 1, 2, 1], dims=[16, 32, 64, 128], decoder_conv_blocks=1,
              in_stack_depth=5, pretraining=False)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 1, 5, 64, 64, generator=g).cuda()
    t = torch.rand(2, 2, 5, 64, 64, generator=g).cuda()
    grads = []
    for frozen in (True, False):
        torch.manual_seed(0)
        vs = FcmaeUNet(model_config=kw, loss_function=MixedLoss(0.5, 0.5, 0.0), lr=1e-3, freeze_encoder=frozen).cuda()
        vs.model.compute_dtype = torch.float32
        opt = vs.configure_optimizers(t_total=4)
        before = {n: p.detach().clone() for n, p in vs.model.named_parameters()}
        opt.zero_grad()
        loss = vs.training_step({"source": x, "target": t}, 0)
        loss.backward()
        grads.append({n: vs.model.engine().g(p).clone() for n, p in vs.model.named_parameters() if not n.startswith("encoder.stem.conv2d")})
        opt.step()
        torch.cuda.synchronize()
        moved = {n: not torch.equal(before[n], p.detach()) for n, p in vs.model.named_parameters() if not n.startswith("encoder.stem.conv2d")}
        if frozen:
            assert opt.n_active < vs.model.engine().flat.numel()
            assert not any(v for n, v in moved.items() if n.startswith("encoder.")), [n for n, v in moved.items() if v and n.startswith("encoder.")]
            assert all(v for n, v in moved.items() if n.startswith("decoder."))
            assert all(not p.requires_grad for n, p in vs.model.named_parameters() if n.startswith("encoder."))
        else:
            assert all(moved.values())
    for n in grads[0]:
        if n.startswith("decoder."):
            assert relerr(grads[0][n], grads[1][n]) <= 1e-5, n
        else:
            assert grads[0][n].abs().max() == 0


# ------------------------------------------------------------------------------------------------ stochastic depth
@pytest.mark.parametrize("which", ["fcmae", "unext2"])
def test_stochastic_depth_matches_reference_golden_fp32(which):
    """`encoder_drop_path_rate` (0.1 in every published VSCyto3D recipe) / `drop_path_rate`: training-mode forward with the
    reference run's per-sample branch scales injected == the reference's output; gradients vs oracle autograd; eval mode
    has no stochastic depth; without injection the drop statistics follow the rate"""
    from oracle import fcmae_ref
    from viscy_amd.fcmae import FullyConvolutionalMAE
    from viscy_amd.unext2 import UNeXt2

    gold = load_golden("droppath.pt")[which]
    kw = gold["kwargs"]
    if which == "fcmae":
        ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**kw), seed=gold["seed"]).train()
        mine = FullyConvolutionalMAE(**kw)
    else:
        ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=gold["seed"]).train()
        with torch.no_grad():
            ref.head.conv[0].adn.A.weight.fill_(1.0)  # no PReLU kink in the gradient comparison
        mine = UNeXt2(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda().train()
    mine.compute_dtype = torch.float32
    x = torch.randn(gold["x_shape"], generator=torch.Generator().manual_seed(gold["x_seed"]))
    for m, sc in zip([m for m in ref.modules() if isinstance(m, unext2_ref.DropPath)], gold["masks"]):
        m.inject = sc
    eng = mine.engine()
    eng._dp_inject = [s.cuda() for s in gold["masks"]]
    out = mine(x.cuda())
    y = ref(x)
    if which == "fcmae":
        assert relerr(out, gold["y"]) <= 1e-3
    assert relerr(out, y) <= 1e-3
    dout = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    y.backward(dout)
    out.backward(dout.cuda())
    for (name, pr), (n2, pm) in zip(ref.named_parameters(), mine.named_parameters()):
        if pr.grad is None or name == "head.conv.0.conv.bias":
            continue
        assert relerr(pm.grad, pr.grad) <= 2e-3, name
    # device-side draws: branches are dropped at about the configured rate, survivors scaled by 1 / keep
    eng._dp_inject = None
    seen = []
    orig = eng.ops.gemm

    def spy(*a, **k):
        if k.get("rscale") is not None:
            seen.append(k["rscale"].clone())
        return orig(*a, **k)

    eng.ops = type("O", (), {**{n: getattr(eng.ops, n) for n in dir(eng.ops) if not n.startswith("__")}, "gemm": staticmethod(spy)})
    with torch.no_grad():
        for _ in range(20):
            mine(x.cuda())
    eng.ops = __import__("viscy_amd.ops", fromlist=["ops"])
    vals = torch.cat(seen).cpu()
    rate = kw.get("encoder_drop_path_rate", None)
    if rate is not None:
        u = vals.unique().tolist()
        assert len(u) == 2 and u[0] == 0.0 and abs(u[1] - 1 / (1 - rate)) < 1e-5, u
        assert abs((vals == 0).float().mean().item() - rate) < 0.08
    mine.eval()
    ref.eval()
    with torch.no_grad():
        assert relerr(mine(x.cuda()), ref(x)) <= 1e-3


def test_contrastive_encoder_stochastic_depth_modes():
    """drop_path_rate (0.1 in the DynaCLR configs): identity in eval mode, random per-sample drops in training mode"""
    from viscy_amd.contrastive import ContrastiveEncoder

    torch.manual_seed(0)
    kw = dict(in_channels=1, in_stack_depth=5, embedding_dim=64, projection_dim=32, depths=(1, 1, 2, 1), dims=(24, 48, 96, 192))
    a = ContrastiveEncoder("convnextv2_tiny", drop_path_rate=0.5, **kw).cuda()
    b = ContrastiveEncoder("convnextv2_tiny", **kw).cuda()
    b.load_state_dict(a.state_dict())
    assert [round(r, 4) for r in a.cfg["drop_path"]] == [0.0, 0.125, 0.25, 0.375, 0.5]
    x = torch.randn(6, 1, 5, 64, 64).cuda()
    a.eval(); b.eval()
    with torch.no_grad():
        assert torch.equal(a(x)[0], b(x)[0])
        a.train(); b.train()
        e1, e2, eb = a(x)[0], a(x)[0], b(x)[0]
    assert not torch.equal(e1, e2) and not torch.equal(e1, eb)


def test_unext2_single_output_channel_bf16_runs_the_head_in_fp32():
    """1 -> 1 channel virtual staining in bf16: the head's 4-channel planes are not whole bf16 vectors, so the head runs in
    fp32 behind the bf16 trunk; forward / gradients track the fp32 oracle within the bf16 bars"""
    kw = dict(in_channels=1, out_channels=1, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True)
    ref, mine = _pair(kw)
    with torch.no_grad():
        ref.head.conv[0].adn.A.weight.fill_(1.0)
        mine.head.conv[0].adn.A.weight.fill_(1.0)
    mine = mine.cuda()
    mine.compute_dtype = torch.bfloat16
    x = torch.randn(2, 1, 5, 64, 64, generator=torch.Generator().manual_seed(3))
    y = ref(x)
    out = mine(x.cuda())
    assert out.shape == (2, 1, 5, 64, 64)
    torch.testing.assert_close(out.cpu(), y.detach(), rtol=2e-2, atol=0.03 * y.abs().max().item())
    dout = torch.randn(y.shape, generator=torch.Generator().manual_seed(4))
    y.backward(dout)
    out.backward(dout.cuda())
    cos = torch.nn.functional.cosine_similarity(
        torch.cat([p.grad.flatten().cpu() for p in mine.parameters()]), torch.cat([p.grad.flatten() for p in ref.parameters()]), dim=0)
    assert cos > 0.99, cos


def test_atto_backbone_under_bf16_autocast_falls_back_to_fp32_kernels():
    kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_atto")
    ref, mine = _pair(kw)
    x = torch.randn(1, 1, 5, 64, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = mine(x.cuda())
        assert relerr(out, ref(x)) <= 1e-3   # fp32 accuracy: the 60-channel decoder rows cannot be bf16 vectors


def test_large_batch_addressing_beyond_4gb_tensors():
    """B = 640 patches: the 4C-wide activations of the last decoder stage pass 4 GB (2.6 M rows x 896 x 2 B), i.e. 32-bit byte
    offsets would wrap.  Forward: the last samples of the big batch equal a small-batch run of the same samples.  Backward:
    with an output gradient on the last samples only, the parameter gradients equal the small-batch gradients."""
    from viscy_amd.unext2 import UNeXt2

    torch.manual_seed(0)
    m = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True).cuda().eval()
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if "grn" in n_:
                p.normal_(0, 0.1)
    m.compute_dtype = torch.bfloat16
    m.grad_mode = "flat"
    B, k = 640, 3
    g = torch.Generator().manual_seed(1)
    tail = torch.randn(k, 1, 5, 256, 256, generator=g).cuda()
    x = torch.zeros(B, 1, 5, 256, 256, device="cuda")
    x[: B - k].normal_()
    x[B - k:] = tail
    with torch.no_grad():
        y_big = m(x)[B - k:].clone()
        y_small = m(tail)
    torch.testing.assert_close(y_big, y_small, rtol=0, atol=0.02 * y_small.abs().max().item())  # bf16 run-to-run noise only
    # backward
    eng = m.engine()
    dout_tail = torch.randn(y_small.shape, generator=g).cuda()
    eng.flat_grad.zero_()
    ys = m(tail)
    ys.backward(dout_tail)
    g_small = eng.flat_grad.clone()
    eng.flat_grad.zero_()
    yb = m(x)
    dout = torch.zeros_like(yb)
    dout[B - k:] = dout_tail
    yb.backward(dout)
    g_big = eng.flat_grad.clone()
    cos = torch.nn.functional.cosine_similarity(g_big, g_small, dim=0).item()
    assert cos > 0.999, cos
    assert abs(g_big.norm().item() / g_small.norm().item() - 1) < 0.02


def test_edge_cases_of_the_widened_paths():
    """degenerate inputs the reference semantics define: NT-Xent without negatives / with one pair, BatchNorm1d eval on a
    single sample, FCMAE mask ratios that round to zero masked cells or leave a single kept cell"""
    from oracle import fcmae_ref
    from oracle.contrastive_ref import NTXentLoss as RefLoss
    from viscy_amd.contrastive import ContrastiveEncoder, NTXentLoss
    from viscy_amd.fcmae import FullyConvolutionalMAE

    e = torch.randn(6, 16, generator=torch.Generator().manual_seed(0))
    same = torch.zeros(6, dtype=torch.long)
    eg = e.cuda().requires_grad_(True)
    l0 = NTXentLoss(0.2)(eg, same.cuda())                      # no negatives anywhere: zero loss, zero gradient
    l0.backward()
    assert l0.item() == 0.0 and eg.grad.abs().max().item() == 0.0
    pair = torch.tensor([0, 0, 1, 1])                           # two classes of two
    lr = RefLoss(0.5)(e[:4], pair)
    lg = NTXentLoss(0.5)(e[:4].cuda(), pair.cuda())
    assert abs(lg.item() - lr.item()) <= 1e-5 * abs(lr.item())
    enc = ContrastiveEncoder("convnextv2_tiny", in_channels=1, in_stack_depth=5, embedding_dim=32, projection_dim=16,
                             depths=(1, 1, 1, 1), dims=(16, 32, 64, 128)).cuda()
    enc.compute_dtype = torch.float32
    enc.eval()
    with torch.no_grad():
        emb, proj = enc(torch.randn(1, 1, 5, 64, 64).cuda())   # eval-mode BatchNorm works on a single sample
    assert emb.shape == (1, 128) and proj.shape == (1, 16) and torch.isfinite(proj).all()
    enc.train()
    with pytest.raises(RuntimeError, match="more than 1 value per channel"):
        enc(torch.randn(1, 1, 5, 64, 64).cuda())                # ... training mode does not (as nn.BatchNorm1d)
    kw = dict(in_channels=1, out_channels=1, encoder_blocks=[1, 1, 1, 1], dims=[16, 32, 64, 128], decoder_conv_blocks=1,
              in_stack_depth=5, pretraining=True)
    ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**kw), seed=2)
    mine = FullyConvolutionalMAE(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    mine.compute_dtype = torch.float32
    x = torch.randn(2, 1, 5, 64, 96, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        y, mask = mine(x.cuda(), mask_ratio=0.1)                # int(6 * 0.1) = 0 masked cells: every token kept
        assert not mask.any() and relerr(y, ref(x, mask=torch.zeros(2, 1, 2, 3, dtype=torch.bool))[0]) <= 1e-3
        low = torch.ones(2, 1, 2, 3, dtype=torch.bool)
        low[0, 0, 0, 1] = False
        low[1, 0, 1, 2] = False                                  # a single kept cell per sample (L = 1 token at stage 3)
        y1, _ = mine(x.cuda(), mask=low.cuda())
        assert relerr(y1, ref(x, mask=low)[0]) <= 1e-3
