"""CPU: the data-parallel training step with the REAL engine schedule (kernels = tests/ref_ops.py), world_size 2 over gloo:
2-rank gradients == 1-rank gradients of the union batch, identical parameters after one AdamW step, steps with several
forwards reduce every bucket exactly once; the device-side optimiser schedule (vsx_adamw_advance restated in ref_ops)
== torch AdamW + MONAI WarmupCosine (LambdaLR); ICNR initialisation (reference blocks.py:14-51)."""

import math
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import ref_ops
from viscy_amd.engine_unext2 import Engine, unext2_apply
from viscy_amd.optim import FlatAdamW, warmup_cosine_lambda
from viscy_amd.parallel import FlatDataParallel
from viscy_amd.step import TrainStep
from viscy_amd.unext2 import UNeXt2

KW = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_atto", head_pool=True)


def _model(seed=0):
    torch.manual_seed(seed)
    m = UNeXt2(**KW)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ".grn." in n:
                p.copy_(torch.randn(p.shape) * 0.1)
    m.compute_dtype, m.grad_mode = torch.float32, "flat"
    m._engine = Engine(m, ops=ref_ops)  # the schedule under test, kernels stated in plain PyTorch
    return m


def _batch(n, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((n, 1, 5, 64, 64), generator=g), torch.randn((n, 2, 5, 64, 64), generator=g)


def _dp_worker(rank, world, init_file, out, mode):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(2)
    m = _model(seed=rank)  # different initial weights per rank: the broadcast must win
    eng = m.engine()
    opt = FlatAdamW(eng, lr=1e-3, ops=ref_ops)
    ddp = FlatDataParallel(eng, opt)
    x, t = _batch(4)
    xs, ts = x[rank::world].contiguous(), t[rank::world].contiguous()
    crit = torch.nn.MSELoss()
    calls = []
    real = ddp.reduce_bucket

    def counting(i):
        if i not in ddp._reduced:
            calls.append(i)
        real(i)

    ddp.reduce_bucket = counting
    if mode == "direct":
        step = TrainStep(m, crit, opt, ddp, use_graph=False)
        loss = step(xs, ts)
    else:  # two forwards, one backward through autograd (CombinedLoader-style step): hooks fire in the LAST backward only
        def loss_fn(a, b):
            return 0.5 * (crit(unext2_apply(m, a[:1]), b[:1]) + crit(unext2_apply(m, a[1:]), b[1:]))

        step = TrainStep(m, None, opt, ddp, use_graph=False, loss_fn=loss_fn)
        loss = step(xs, ts)
    out[rank] = (eng.flat_grad.clone(), eng.flat.clone(), float(loss), list(calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["direct", "two_forwards"])
def test_two_rank_step_equals_one_rank_step_on_the_union_batch(mode):
    world = 2
    init_file = tempfile.mktemp()
    out = mp.Manager().dict()
    mp.spawn(_dp_worker, args=(world, init_file, out, mode), nprocs=world, join=True)
    (g0, p0, l0, c0), (g1, p1, l1, c1) = out[0], out[1]
    assert torch.equal(g0, g1) and torch.equal(p0, p1)   # both ranks hold the reduced gradient and the same parameters
    assert sorted(c0) == [0, 1, 2] and sorted(c1) == [0, 1, 2]  # every bucket reduced exactly once per step
    # one rank, union batch, same initial weights (rank 0's)
    m = _model(seed=0)
    eng = m.engine()
    opt = FlatAdamW(eng, lr=1e-3, ops=ref_ops)
    x, t = _batch(4)
    step = TrainStep(m, torch.nn.MSELoss(), opt, None, use_graph=False)
    loss = step(x, t)
    # SUM over ranks x grad_scale 1/world == gradient of the mean loss over the union batch
    gd = 0.5 * g0
    rel = ((gd - eng.flat_grad).norm() / eng.flat_grad.norm()).item()
    assert rel < 1e-4, rel
    assert abs(0.5 * (l0 + l1) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    # parameters after the AdamW step: Adam's first step is lr * sign-like, so compare the update vectors
    m0 = _model(seed=0)
    init = m0.engine().flat
    du, d1 = p0 - init, eng.flat - init
    assert (du - d1).abs().max().item() <= 2.1e-3  # never more than one full Adam step (lr) apart, each way
    # Adam's first step is lr * g / (|g| + eps): wherever the gradient is not round-off (exactly-zero gradients such as the
    # bias in front of InstanceNorm turn into +-lr by sign noise) the two updates must agree closely
    big = eng.flat_grad.abs() > 1e-5 * eng.flat_grad.abs().max()
    assert big.float().mean().item() > 0.5
    assert (du - d1)[big].abs().max().item() < 2e-5
    assert torch.nn.functional.cosine_similarity(du, d1, dim=0).item() > 0.99


def test_device_side_schedule_matches_torch_adamw_with_warmup_cosine():
    """FlatAdamW's per-step scalars come from device state (step counter + constants): N steps on a fixed gradient sequence
    == torch.optim.AdamW + LambdaLR(WarmupCosine) stepped per batch (viscy_utils/optimizers.py:50-61)."""
    torch.manual_seed(0)

    class _Eng:
        pass

    eng = _Eng()
    n = 1000
    eng.flat = torch.randn(n)
    eng.flat_grad = torch.zeros(n)
    ref_p = torch.nn.Parameter(eng.flat.clone())
    topt = torch.optim.AdamW([ref_p], lr=2e-3)
    total, warm, mult = 12, 3, 1e-3
    sch = torch.optim.lr_scheduler.LambdaLR(topt, lambda s: warmup_cosine_lambda(s, warm, total, mult))
    opt = FlatAdamW(eng, lr=2e-3, schedule="WarmupCosine", warmup_steps=warm, t_total=total, warmup_multiplier=mult, ops=ref_ops)
    g = torch.Generator().manual_seed(1)
    for k in range(total):
        grad = torch.randn(n, generator=g)
        eng.flat_grad.copy_(grad)
        ref_p.grad = grad.clone()
        assert opt.current_lr() == pytest.approx(sch.get_last_lr()[0], rel=1e-12, abs=1e-15)
        opt.step()
        topt.step()
        sch.step()
        assert float(opt.hyper[0]) == pytest.approx(2e-3 * warmup_cosine_lambda(k, warm, total, mult), rel=1e-6, abs=1e-12)
        assert int(opt.step_dev) == opt.t == k + 1
    torch.testing.assert_close(eng.flat, ref_p.detach(), rtol=1e-5, atol=1e-6)
    # checkpoint round trip restores the device counter
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict().items()}
    opt2 = FlatAdamW(eng, lr=2e-3, schedule="WarmupCosine", warmup_steps=warm, t_total=total, warmup_multiplier=mult, ops=ref_ops)
    opt2.load_state_dict(sd)
    assert int(opt2.step_dev) == total and opt2.t == total


def test_icnr_initialisation_makes_subkernels_identical():
    """reference blocks.py:14-51 (``icnr_init``): the out-channel groups that one pixel-shuffle output channel is assembled
    from start as copies of one kernel, so the shuffle starts as nearest-neighbour up-sampling (no checkerboard)."""
    from viscy_amd.unext2 import _icnr_

    torch.manual_seed(0)
    for shape in [(64, 16), (32, 8, 1, 1), (8, 32, 1, 1, 1)]:
        w = torch.empty(shape)
        _icnr_(w, 2)
        oc = shape[0]
        groups = w.reshape(oc // 4, 4, -1)
        assert torch.equal(groups, groups[:, :1].expand_as(groups)), shape
        assert w.std() > 0  # not degenerate
        # what that buys: pixel_shuffle(conv1x1(x)) is piecewise constant over each 2x2 output cell
        if len(shape) == 4:
            x = torch.randn(1, shape[1], 5, 5)
            y = torch.nn.functional.pixel_shuffle(torch.nn.functional.conv2d(x, w), 2)
            cells = y.unfold(2, 2, 2).unfold(3, 2, 2)
            assert torch.allclose(cells, cells[..., :1, :1].expand_as(cells))
    # every decoder stage's last fc2 and the head's 1x1x1 convolution are initialised that way in the model
    m = UNeXt2(**KW)
    for st in m.decoder.decoder_stages:
        w = st.conv.blocks[-1].mlp.fc2.weight
        gq = w.detach().reshape(w.shape[0] // 4, 4, -1)
        assert torch.equal(gq, gq[:, :1].expand_as(gq))
    w = m.head.conv[1].weight.detach()
    gq = w.reshape(w.shape[0] // 4, 4, -1)
    assert torch.equal(gq, gq[:, :1].expand_as(gq))
