"""CPU: the hand-written forward/backward *schedule* of viscy_amd.engine_unext2 (weight folding,
GRN / InstanceNorm backward algebra, layouts, gather GEMMs) equals autograd of the oracle model,
with the kernels replaced by their plain-PyTorch statements (tests/ref_ops.py)."""

import pytest
import torch

from oracle import unext2_ref
from tests import ref_ops
from viscy_amd.engine_unext2 import Engine
from viscy_amd.unext2 import UNeXt2


def _pair(kw, seed=7):
    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=seed)
    mine = UNeXt2(**kw)
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    for (k1, v1), (k2, v2) in zip(mine.state_dict().items(), ref.state_dict().items()):
        assert v1.shape == v2.shape, (k1, v1.shape, v2.shape)
    mine.load_state_dict(ref.state_dict(), strict=True)
    return ref, mine


CASES = [
    ("atto_pool", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_atto", head_pool=True), (2, 64, 96)),
    ("femto_z15", dict(in_channels=2, out_channels=2, in_stack_depth=15, out_stack_depth=5, backbone="convnextv2_femto"), (1, 64, 64)),
    ("tiny_nopool", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny"), (1, 64, 64)),
    # decoder_upsample_pre_conv=True: MONAI SubpixelUpsample's 3x3 convolution in front of every pixel shuffle (blocks.py:138-146)
    ("femto_preconv", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_femto",
                           decoder_upsample_pre_conv=True), (2, 64, 96)),
]


@pytest.mark.parametrize("tag,kw,bhw", CASES, ids=[c[0] for c in CASES])
def test_schedule_matches_oracle_autograd(tag, kw, bhw):
    torch.manual_seed(0)
    ref, mine = _pair(kw)
    B, H, W = bhw
    x = torch.randn(B, kw["in_channels"], kw["in_stack_depth"], H, W)
    eng = Engine(mine, ops=ref_ops)
    with torch.no_grad():
        out, sv = eng.forward(x, torch.float32, need_bwd=True)
    y = ref(x)
    assert out.shape == y.shape
    torch.testing.assert_close(out, y.detach(), rtol=2e-4, atol=2e-5)
    dout = torch.randn_like(y)
    y.backward(dout)
    with torch.no_grad():
        eng.backward(sv, dout)
    worst = 0.0
    for (name, p_ref), p in zip(ref.named_parameters(), mine.parameters()):
        g = eng.g(p)
        denom = p_ref.grad.abs().max().clamp_min(1e-6)
        if name == "head.conv.0.conv.bias":
            # a bias in front of InstanceNorm has an exactly-zero gradient; both sides are round-off
            assert g.abs().max() < 1e-3 and p_ref.grad.abs().max() < 1e-3
            continue
        err = ((g - p_ref.grad).abs().max() / denom).item()
        worst = max(worst, err)
        assert err < 2e-3, (name, err)
    print(tag, "max rel grad err", worst)


def test_state_dict_compat():
    m = UNeXt2(backbone="convnextv2_atto")
    sd = m.state_dict()
    assert len(sd) == 213  # reference tests/test_state_dict_compat.py:35
    assert {k.split(".")[0] for k in sd} == {"decoder", "encoder_stages", "head", "stem"}
    for key in ["stem.conv.weight", "encoder_stages.stages_1.blocks.1.mlp.fc2.bias",
                "decoder.decoder_stages.0.conv.blocks.0.conv_dw.weight", "decoder.decoder_stages.2.conv.blocks.0.mlp.grn.bias",
                "head.conv.1.weight", "head.conv.0.adn.A.weight"]:
        assert key in sd
    assert m.num_blocks == 6 and m.out_stack_depth == 5
    # decoder_upsample_pre_conv=True adds one Conv2d(C, C, 3) per decoder stage under MONAI's module names, ICNR-initialised
    # (identical 2x2 sub-kernels), and leaves the stage's last fc2 on timm's init (blocks.py:147)
    ref = unext2_ref.UNeXt2(backbone="convnextv2_atto", decoder_upsample_pre_conv=True)
    mp = UNeXt2(backbone="convnextv2_atto", decoder_upsample_pre_conv=True)
    assert list(mp.state_dict()) == list(ref.state_dict()) and len(mp.state_dict()) == 213 + 6
    assert [tuple(v.shape) for v in mp.state_dict().values()] == [tuple(v.shape) for v in ref.state_dict().values()]
    w = mp.state_dict()["decoder.decoder_stages.1.upsample.pixelshuffle.conv_block.weight"]
    assert w.shape == (160, 160, 3, 3)
    groups = w.view(40, 4, 160, 3, 3)
    assert torch.equal(groups[:, 0], groups[:, 3]) and not torch.equal(groups[0, 0], groups[1, 0])
    fc2 = mp.state_dict()["decoder.decoder_stages.1.conv.blocks.1.mlp.fc2.weight"]
    assert not torch.equal(fc2[0], fc2[1])  # no ICNR on the stage when the pre-convolution carries it


def test_bad_depth_and_cpu_forward_raise():
    with pytest.raises(ValueError, match="not divisible"):
        UNeXt2(in_stack_depth=7)
    m = UNeXt2(backbone="convnextv2_atto")
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 1, 5, 64, 64))


# ---------------------------------------------------------------- FCMAE masked pre-training schedule
def _fcmae_masked_case(tag):
    from oracle import fcmae_ref
    from tests.conftest import load_golden
    from viscy_amd.fcmae import FullyConvolutionalMAE

    gold = load_golden("fcmae_masked.pt")[tag]
    kw = gold["kwargs"]
    ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**kw), seed=gold["seed"])
    mine = FullyConvolutionalMAE(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    g = torch.Generator().manual_seed(gold["x_seed"])
    x = torch.randn(gold["x_shape"], generator=g)
    return gold, ref, mine, x


@pytest.mark.parametrize("tag", ["small_z5_r50", "two_ch_r75"])
def test_row_maps_are_the_boolean_index_order(tag):
    from oracle import fcmae_ref
    from viscy_amd.fcmae import stage_row_maps

    gold, _, _, x = _fcmae_masked_case(tag)
    low = gold["mask_low"]
    kept = int((~low).flatten(1).sum(1)[0])
    B, H, W = x.shape[0], x.shape[-2] // 4, x.shape[-1] // 4
    maps = stage_row_maps(~low, [(H >> i, W >> i) for i in range(4)], kept)
    for i, (idx, inv, keep, L) in enumerate(maps):
        h, w = H >> i, W >> i
        u = fcmae_ref.upsample_mask(~low, (B, 1, h, w))[:, 0]
        feat = torch.randn(B, 3, h, w)
        tok = fcmae_ref._tokens(feat, u.unsqueeze(1)).reshape(-1, 3)      # reference order (boolean indexing)
        rows = feat.permute(0, 2, 3, 1).reshape(-1, 3)
        assert L * B == tok.shape[0] == idx.numel()
        assert torch.equal(rows[idx.long()], tok)
        assert torch.equal(ref_ops.rows_select(tok, inv, B * h * w, 3).view(B, h, w, 3).permute(0, 3, 1, 2),
                           fcmae_ref._untokens(tok.view(B, -1, 3), feat.shape, u.unsqueeze(1)))
        assert torch.equal(ref_ops.rows_select(rows, keep, B * h * w, 3), rows * u.reshape(-1, 1))


@pytest.mark.parametrize("tag", ["small_z5_r50", "two_ch_r75"])
def test_masked_schedule_matches_reference_golden(tag):
    """forward + every parameter gradient of the masked path: engine schedule (kernels = plain torch) vs the reference's own
    run stored in tests/golden/fcmae_masked.pt and vs autograd of the oracle."""
    from oracle import fcmae_ref
    from viscy_amd.fcmae import stage_row_maps

    gold, ref, mine, x = _fcmae_masked_case(tag)
    low = gold["mask_low"]
    kept = int((~low).flatten(1).sum(1)[0])
    H, W = x.shape[-2] // 4, x.shape[-1] // 4
    masks = stage_row_maps(~low, [(H >> i, W >> i) for i in range(4)], kept)
    eng = Engine(mine._core, ops=ref_ops)
    with torch.no_grad():
        out, sv = eng.forward(x, torch.float32, need_bwd=True, masks=masks)
    torch.testing.assert_close(out, gold["y"], rtol=2e-4, atol=1e-4 * gold["y"].abs().max().item())
    y, mask = ref(x, mask=low)
    loss = fcmae_ref.MaskedMSELoss()(y, x, mask)
    assert abs(loss.item() - gold["loss"]) < 1e-6 * max(1.0, abs(gold["loss"]))
    (dy,) = torch.autograd.grad(loss, y, retain_graph=True)
    loss.backward()
    with torch.no_grad():
        eng.backward(sv, dy)
    mine_named = dict(mine.named_parameters())
    worst = 0.0
    for name, p_ref in ref.named_parameters():
        if name.startswith("encoder.stem.conv2d"):
            continue
        gr = eng.g(mine_named[name])
        err = ((gr - p_ref.grad).abs().max() / p_ref.grad.abs().max().clamp_min(1e-6)).item()
        worst = max(worst, err)
        assert err < 2e-3, (name, err)
    for name, gg in gold["grads"].items():
        gr = eng.g(mine_named[name])
        assert ((gr - gg).abs().max() / gg.abs().max().clamp_min(1e-6)).item() < 2e-3, name
    print(tag, "masked max rel grad err", worst)


def test_fcmae_2d_model_runs_on_its_conv2d_stem():
    """in_stack_depth = 1 (a 2-D FCMAE): the reference takes the conv2d branch whenever x.shape[2] == 1 (fcmae.py:369-370),
    so the trained stem of such a checkpoint is conv2d — forward == oracle, the gradient lands on conv2d, conv3d gets none
    (the engine used to run these models on the untrained conv3d weights)"""
    from oracle import fcmae_ref
    from viscy_amd.fcmae import FullyConvolutionalMAE

    kw = dict(in_channels=2, out_channels=2, encoder_blocks=[1, 1, 1, 1], dims=[16, 32, 64, 128], decoder_conv_blocks=1,
              stem_kernel_size=(1, 4, 4), in_stack_depth=1)
    ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**kw), seed=11)
    mine = FullyConvolutionalMAE(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(2, 2, 1, 64, 64, generator=torch.Generator().manual_seed(5))
    eng = Engine(mine._core, ops=ref_ops)
    with torch.no_grad():
        out, sv = eng.forward(x, torch.float32, need_bwd=True)
    y, _ = ref(x)
    torch.testing.assert_close(out, y, rtol=2e-4, atol=1e-4 * y.abs().max().item())
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(6))
    y.backward(dy)
    with torch.no_grad():
        eng.backward(sv, dy)
    named = dict(mine.named_parameters())
    g2 = eng.g(named["encoder.stem.conv2d.weight"])
    gr = dict(ref.named_parameters())["encoder.stem.conv2d.weight"].grad
    assert ((g2 - gr).abs().max() / gr.abs().max()).item() < 2e-3
    assert float(eng.g(named["encoder.stem.conv3d.weight"]).abs().max()) == 0.0
    # and the forward really reads conv2d: perturbing it changes the output, perturbing conv3d does not
    with torch.no_grad():
        named["encoder.stem.conv3d.weight"].add_(1.0)
        out3, _ = eng.forward(x, torch.float32, need_bwd=False)
        named["encoder.stem.conv2d.weight"].add_(1.0)
        out2, _ = eng.forward(x, torch.float32, need_bwd=False)
    assert torch.equal(out3, out) and not torch.equal(out2, out)


def test_inference_schedule_skips_preactivation_store():
    """need_bwd=False: fc1 keeps only gelu(h) (C = NULL); same output as the training-mode forward"""
    torch.manual_seed(0)
    ref, mine = _pair(CASES[0][1])
    x = torch.randn(1, 1, 5, 64, 64)
    eng = Engine(mine, ops=ref_ops)
    with torch.no_grad():
        a, sv = eng.forward(x, torch.float32, need_bwd=False)
        b, _ = eng.forward(x, torch.float32, need_bwd=True)
    assert sv is None and torch.equal(a, b)


# ---------------------------------------------------------------- DynaCLR ContrastiveEncoder schedule
@pytest.mark.parametrize("tag", ["v2_small_z9", "v1_small_z5"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_contrastive_schedule_matches_reference_golden(mode, tag):
    """embedding / projection + every parameter gradient + BatchNorm running statistics: engine schedule (kernels = plain
    torch) vs the reference-generated golden (tests/golden/contrastive.pt, G10) and vs autograd of the oracle"""
    from oracle import contrastive_ref as C
    from tests.conftest import load_golden
    from viscy_amd.contrastive import ContrastiveEncoder

    gold = load_golden("contrastive.pt")[tag]
    ref = C.randomize_encoder_(C.ContrastiveEncoder(**gold["kwargs"], **gold["arch"]), seed=gold["seed"])
    mine = ContrastiveEncoder(**gold["kwargs"], **gold["arch"])
    assert list(mine.state_dict().keys()) == gold["keys"]
    mine.load_state_dict(ref.state_dict(), strict=True)
    getattr(ref, mode)()
    getattr(mine, mode)()
    x = torch.randn(gold["x_shape"], generator=torch.Generator().manual_seed(gold["x_seed"]))
    eng = Engine(mine._core, ops=ref_ops)
    with torch.no_grad():
        (emb, proj), sv = eng.forward(x, torch.float32, need_bwd=True)
    sc = lambda t: 1e-4 * t.abs().max().item()  # noqa: E731
    torch.testing.assert_close(emb, gold[mode][0], rtol=2e-4, atol=sc(gold[mode][0]))
    torch.testing.assert_close(proj, gold[mode][1], rtol=2e-4, atol=sc(gold[mode][1]))
    if mode == "train":
        for k, v in gold["running_after"].items():
            torch.testing.assert_close(mine.state_dict()[k].float(), v.float(), rtol=1e-4, atol=1e-5)
    er, pr = ref(x)
    g = torch.Generator().manual_seed(3)
    de, dp = torch.randn(er.shape, generator=g), torch.randn(pr.shape, generator=g)
    ((er * de).sum() + (pr * dp).sum()).backward()
    with torch.no_grad():
        eng.backward(sv, (de, dp))
    named = dict(mine.named_parameters())
    worst = 0.0
    for name, p_ref in ref.named_parameters():
        gr = eng.g(named[name])
        if mode == "train" and name in ("projection.0.bias", "projection.3.bias"):
            # a bias in front of a train-mode BatchNorm has an exactly-zero gradient; both sides are round-off
            assert gr.abs().max() < 1e-4 and p_ref.grad.abs().max() < 1e-4
            continue
        err = ((gr - p_ref.grad).abs().max() / p_ref.grad.abs().max().clamp_min(1e-6)).item()
        worst = max(worst, err)
        assert err < 2e-3, (name, err)
    print(tag, mode, "contrastive max rel grad err", worst)


def test_contrastive_paired_forward_equals_two_calls():
    """one trunk pass over [anchor; positive] with per-group BatchNorm == two separate forwards of the oracle (what the
    reference's training_step does): projections, running statistics after both calls, every parameter gradient"""
    from oracle import contrastive_ref as C
    from tests.conftest import load_golden
    from viscy_amd.contrastive import ContrastiveEncoder

    gold = load_golden("contrastive.pt")["v1_small_z5"]
    ref = C.randomize_encoder_(C.ContrastiveEncoder(**gold["kwargs"], **gold["arch"]), seed=gold["seed"]).train()
    mine = ContrastiveEncoder(**gold["kwargs"], **gold["arch"])
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.train()
    g = torch.Generator().manual_seed(8)
    a = torch.randn(3, 1, 5, 64, 64, generator=g)
    p = a + 0.3 * torch.randn(a.shape, generator=g)
    eng = Engine(mine._core, ops=ref_ops)
    with torch.no_grad():
        (emb, proj), sv = eng.forward(torch.cat((a, p)), torch.float32, need_bwd=True, bn_groups=2)
    ea, pa = ref(a)
    ep, pp = ref(p)
    torch.testing.assert_close(proj, torch.cat((pa, pp)).detach(), rtol=2e-4, atol=1e-4 * pa.abs().max().item())
    torch.testing.assert_close(emb, torch.cat((ea, ep)).detach(), rtol=2e-4, atol=1e-4 * ea.abs().max().item())
    for k, v in ref.state_dict().items():
        if "running" in k or "num_batches" in k:
            torch.testing.assert_close(mine.state_dict()[k].float(), v.float(), rtol=1e-4, atol=1e-5)
    loss = C.NTXentLoss(temperature=0.3)(torch.cat((pa, pp)), torch.cat((torch.arange(3), torch.arange(3))))
    dpa, dpp = torch.autograd.grad(loss, (pa, pp), retain_graph=True)
    loss.backward()
    with torch.no_grad():
        eng.backward(sv, (None, torch.cat((dpa, dpp))))
    named = dict(mine.named_parameters())
    for name, p_ref in ref.named_parameters():
        gr = eng.g(named[name])
        if name in ("projection.0.bias", "projection.3.bias", "encoder.head.norm.bias"):
            # a per-batch constant shift in front of a train-mode BatchNorm: exactly-zero gradient, both sides are round-off
            assert gr.abs().max() < 1e-4 and p_ref.grad.abs().max() < 1e-4
            continue
        err = ((gr - p_ref.grad).abs().max() / p_ref.grad.abs().max().clamp_min(1e-6)).item()
        assert err < 2e-3, (name, err)


# ---------------------------------------------------------------- stochastic depth (drop_path_rate / encoder_drop_path_rate)
@pytest.mark.parametrize("which", ["fcmae", "unext2"])
def test_stochastic_depth_schedule_matches_reference_golden(which):
    """training-mode forward with the reference run's per-sample branch scales injected == the reference's output
    (tests/golden/droppath.pt), and the backward == oracle autograd with the same scales"""
    from oracle import fcmae_ref
    from tests.conftest import load_golden
    from viscy_amd.fcmae import FullyConvolutionalMAE

    gold = load_golden("droppath.pt")[which]
    kw = gold["kwargs"]
    if which == "fcmae":
        ref = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**kw), seed=gold["seed"]).train()
        mine = FullyConvolutionalMAE(**kw)
        core = mine._core
    else:
        ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=gold["seed"]).train()
        mine = core = UNeXt2(**kw)
        assert [round(r, 6) for r in core.cfg["drop_path"][1:]] == [round(r, 6) for r in gold["rates"]]
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine.train()
    x = torch.randn(gold["x_shape"], generator=torch.Generator().manual_seed(gold["x_seed"]))
    for m, sc in zip([m for m in ref.modules() if isinstance(m, unext2_ref.DropPath)], gold["masks"]):
        m.inject = sc
    eng = Engine(core, ops=ref_ops)
    eng._dp_inject = [s.clone() for s in gold["masks"]]
    with torch.no_grad():
        out, sv = eng.forward(x, torch.float32, need_bwd=True)
    assert eng._dp_inject == []
    torch.testing.assert_close(out, gold["y"], rtol=2e-4, atol=1e-4 * gold["y"].abs().max().item())
    y = ref(x)
    dout = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    y.backward(dout)
    with torch.no_grad():
        eng.backward(sv, dout)
    named = dict(mine.named_parameters())
    for name, p_ref in ref.named_parameters():
        if p_ref.grad is None or name == "head.conv.0.conv.bias":
            continue
        gr = eng.g(named[name])
        err = ((gr - p_ref.grad).abs().max() / p_ref.grad.abs().max().clamp_min(1e-6)).item()
        assert err < 2e-3, (name, err)
    # evaluation mode: no stochastic depth
    mine.eval()
    ref.eval()
    with torch.no_grad():
        out_e, _ = eng.forward(x, torch.float32, need_bwd=False)
        torch.testing.assert_close(out_e, ref(x), rtol=2e-4, atol=1e-4 * gold["y"].abs().max().item())
