"""CPU: the oracle (oracle/*.py) against the committed golden fixtures, which were generated
from the reference itself by oracle/validate_against_reference.py, plus the structural pins
the reference's own tests hold (tests/test_state_dict_compat.py:33-55, test_unext2.py)."""

import pytest
import torch

from oracle import loss_ref, transforms_ref, unext2_ref
from tests.conftest import load_golden


def test_state_dict_compat_atto():
    m = unext2_ref.UNeXt2(backbone="convnextv2_atto")
    sd = m.state_dict()
    assert len(sd) == 213
    assert {k.split(".")[0] for k in sd} == {"decoder", "encoder_stages", "head", "stem"}
    for key in [
        "stem.conv.weight",
        "stem.conv.bias",
        "encoder_stages.stages_1.blocks.1.mlp.fc2.bias",
        "decoder.decoder_stages.0.conv.blocks.0.conv_dw.weight",
        "decoder.decoder_stages.0.conv.blocks.0.mlp.fc1.bias",
        "decoder.decoder_stages.2.conv.blocks.0.mlp.grn.bias",
        "head.conv.1.weight",
    ]:
        assert key in sd


def test_tiny_param_count_and_shapes():
    m = unext2_ref.UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True)
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 32.04e6) < 0.05e6  # 32.04 M
    assert len(m.state_dict()) == 273
    x = torch.randn(1, 1, 5, 64, 64)
    with torch.no_grad():
        assert m(x).shape == (1, 2, 5, 64, 64)


def test_bad_depth_raises():
    with pytest.raises(ValueError, match="not divisible"):
        unext2_ref.UNeXt2(in_stack_depth=7)


@pytest.mark.parametrize("tag", ["atto_pool", "femto_z15", "tiny_pool", "femto_preconv"])
def test_forward_golden(tag):
    g = load_golden("unext2_forward.pt")[tag]
    m = unext2_ref.randomize_(unext2_ref.UNeXt2(**g["kwargs"]), seed=g["seed"]).eval()
    assert len(m.state_dict()) == g["n_keys"]
    cs = sum(p.double().sum() for p in m.parameters()).item()
    assert abs(cs - g["param_checksum"]) <= 1e-6 * max(1.0, abs(g["param_checksum"]))
    x = torch.randn(g["x_shape"], generator=torch.Generator().manual_seed(g["x_seed"]))
    with torch.no_grad():
        y = m(x)
    torch.testing.assert_close(y, g["y"], rtol=1e-5, atol=1e-6)


def test_stem_golden():
    g = load_golden("stem.pt")
    s = unext2_ref.UNeXt2Stem(1, 96, (5, 4, 4), 5)
    s.load_state_dict({"conv.weight": g["weight"], "conv.bias": g["bias"]})
    with torch.no_grad():
        torch.testing.assert_close(s(g["x"]), g["y"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["rand_192", "corr_192", "corr_256", "corr_176x208"])
def test_loss_golden(tag):
    c = load_golden("loss.pt")[tag]
    gen = torch.Generator().manual_seed(c["seed"])
    target = torch.rand(c["shape"], generator=gen)
    pred = target + 0.1 * torch.randn(c["shape"], generator=gen) if c["corr"] else torch.rand(c["shape"], generator=gen)
    pred.requires_grad_(True)
    ms = loss_ref.ms_ssim_25d(pred, target, clamp=True)
    torch.testing.assert_close(ms, c["ms_ssim"], rtol=1e-5, atol=1e-7)
    loss = loss_ref.mixed_loss(pred, target, 0.5, 0.0, 0.5)
    torch.testing.assert_close(loss, c["loss"], rtol=1e-5, atol=1e-7)
    loss.backward()
    sample = pred.grad.flatten()[:: max(1, pred.grad.numel() // 4096)]
    torch.testing.assert_close(sample, c["grad_sample"], rtol=1e-4, atol=1e-9)
    # L1-only branch is bit-exact vs F.l1_loss (reference test_mixed_loss.py:85-99)
    assert torch.equal(loss_ref.mixed_loss(pred.detach(), target, 1.0, 0, 0), torch.nn.functional.l1_loss(pred.detach(), target))


def test_mixed_loss_all_zero_raises():
    with pytest.raises(ValueError):
        loss_ref.mixed_loss(torch.zeros(1), torch.zeros(1), 0, 0, 0)


def test_normalize_golden():
    g = load_golden("normalize.pt")
    assert torch.equal(transforms_ref.normalize_sampled(g["x"], g["mean"], g["std"]), g["y"])
    out = transforms_ref.normalize_sampled(g["kat_in"], torch.tensor(60.0), torch.tensor(10.0))
    torch.testing.assert_close(out, (g["kat_in"] - 60.0) / (10.0 + 1e-8))  # test_normalize.py:49-67
    assert torch.equal(out, g["kat_out"])
    assert torch.equal(transforms_ref.minmax_sampled(g["kat_in"], torch.tensor(55.0), torch.tensor(65.0)), g["mm_out"])
    # clipping KAT (test_normalize.py:96-102)
    x = torch.full((1, 1, 8, 64, 64), 200.0)
    assert torch.allclose(transforms_ref.minmax_sampled(x, torch.tensor(5.0), torch.tensor(95.0)), torch.ones_like(x))


# ------------------------------------------------------------------------------------------------ FCMAE dense path (§8 f2)
@pytest.mark.parametrize("tag", ["small_z5", "vscyto3d_z15", "head_conv_z5"])
def test_fcmae_forward_golden_and_key_compat(tag):
    """oracle/fcmae_ref.py reproduces what the REFERENCE's fcmae.py produced (validate_against_reference.py::g9_fcmae), and
    the MI355X parameter holder exposes exactly the reference's state-dict (keys, order, shapes) and loads it strictly."""
    from oracle import fcmae_ref
    from viscy_amd.fcmae import FullyConvolutionalMAE

    g = load_golden("fcmae_forward.pt")[tag]
    m = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**g["kwargs"]), seed=g["seed"]).eval()
    assert list(m.state_dict().keys()) == g["keys"] and len(g["keys"]) == g["n_keys"]
    cs = sum(p.double().sum() for p in m.parameters()).item()
    assert abs(cs - g["param_checksum"]) <= 1e-6 * max(1.0, abs(g["param_checksum"]))
    x = torch.randn(g["x_shape"], generator=torch.Generator().manual_seed(g["x_seed"]))
    with torch.no_grad():
        y = m(x)
    torch.testing.assert_close(y, g["y"], rtol=1e-5, atol=1e-6)
    mine = FullyConvolutionalMAE(**g["kwargs"])
    sd = mine.state_dict()
    assert list(sd.keys()) == g["keys"]
    assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in m.state_dict().values()]
    mine.load_state_dict(m.state_dict(), strict=True)
    assert mine.out_stack_depth == m.out_stack_depth and mine.num_blocks == m.num_blocks
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        mine(x)
    with pytest.raises(RuntimeError, match="HIP kernels only"):  # the masked path has no CPU fallback either
        mine(x, mask_ratio=0.5)
    with pytest.raises(ValueError, match="every cell"):
        mine(x, mask_ratio=1.0)


@pytest.mark.parametrize("tag", ["small_z5_r50", "two_ch_r75"])
def test_fcmae_masked_oracle_matches_reference_golden(tag):
    """masked pre-training path: the oracle fed the reference's own generate_mask draw reproduces the reference's output,
    MaskedMSELoss value and gradients stored by oracle/validate_against_reference.py (G9)."""
    from oracle import fcmae_ref, unext2_ref

    g = load_golden("fcmae_masked.pt")[tag]
    m = unext2_ref.randomize_(fcmae_ref.FullyConvolutionalMAE(**g["kwargs"]), seed=g["seed"])
    x = torch.randn(g["x_shape"], generator=torch.Generator().manual_seed(g["x_seed"]))
    y, mask = m(x, mask=g["mask_low"])
    torch.testing.assert_close(y, g["y"], rtol=1e-5, atol=1e-6)
    assert abs(mask.float().mean().item() - g["mask_ratio"]) < 1e-6
    loss = fcmae_ref.MaskedMSELoss()(y, x, mask)
    assert abs(loss.item() - g["loss"]) <= 1e-6 * abs(g["loss"])
    loss.backward()
    named = dict(m.named_parameters())
    for name, gg in g["grads"].items():
        torch.testing.assert_close(named[name].grad, gg, rtol=1e-4, atol=1e-7 + 1e-4 * gg.abs().max().item())
    # generate_mask: exactly int(n * ratio) cells per sample
    torch.manual_seed(0)
    low = fcmae_ref.generate_mask((3, 1, 5, 128, 160), 32, 0.6)
    assert low.shape == (3, 1, 4, 5) and low.flatten(1).sum(1).tolist() == [12, 12, 12]


# ------------------------------------------------------------------------------------------------ DynaCLR path (§8 f3)
@pytest.mark.parametrize("tag", ["v1_tiny_z15", "v2_small_z9", "v1_small_z5"])
def test_contrastive_encoder_golden(tag):
    """oracle ContrastiveEncoder == what the REFERENCE's encoder.py produced on a stub timm (G10): eval + train outputs,
    BatchNorm running statistics after the train-mode call, state-dict keys"""
    from oracle import contrastive_ref as C

    g = load_golden("contrastive.pt")[tag]
    m = C.randomize_encoder_(C.ContrastiveEncoder(**g["kwargs"], **g["arch"]), seed=g["seed"])
    assert list(m.state_dict().keys()) == g["keys"]
    x = torch.randn(g["x_shape"], generator=torch.Generator().manual_seed(g["x_seed"]))
    m.eval()
    with torch.no_grad():
        e, p = m(x)
    torch.testing.assert_close(e, g["eval"][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(p, g["eval"][1], rtol=1e-5, atol=1e-6)
    m.train()
    with torch.no_grad():
        e, p = m(x)
    torch.testing.assert_close(p, g["train"][1], rtol=1e-5, atol=1e-5)
    for k, v in g["running_after"].items():
        torch.testing.assert_close(m.state_dict()[k], v, rtol=1e-5, atol=1e-6)
    if tag == "v1_tiny_z15":  # reference test_encoder.py:8-22 shapes
        assert e.shape == (4, 768) and p.shape == (4, 128)


def test_ntxent_golden_and_simclr_cross_entropy():
    """NTXentHCL values / gradients stored from the REFERENCE's loss.py (G10); beta = 0 equals the textbook SimCLR
    cross-entropy over each row's similarities without the self term (the pml semantics the reference relies on)"""
    from oracle import contrastive_ref as C

    cases = load_golden("contrastive.pt")["loss"]
    for c in cases.values():
        e = torch.randn(2 * c["n"], c["dim"], generator=torch.Generator().manual_seed(c["seed"]), requires_grad=True)
        labels = torch.cat((torch.arange(c["n"]), torch.arange(c["n"])))
        loss = C.NTXentHCL(temperature=c["temperature"], beta=c["beta"])(e, labels)
        assert abs(loss.item() - c["loss"]) <= 1e-6 * abs(c["loss"])
        (gr,) = torch.autograd.grad(loss, e)
        torch.testing.assert_close(gr, c["grad"], rtol=1e-5, atol=1e-8)
        if c["beta"] == 0.0:
            en = torch.nn.functional.normalize(e.detach(), dim=1)
            s = en @ en.t() / c["temperature"]
            s.fill_diagonal_(float("-inf"))
            n = c["n"]
            tgt = torch.cat((torch.arange(n) + n, torch.arange(n)))
            assert abs(torch.nn.functional.cross_entropy(s, tgt).item() - c["loss"]) <= 1e-5 * abs(c["loss"])
    # reference test_loss.py: a stem that cannot fold the depth is rejected with the reference's message
    with pytest.raises(ValueError, match="more channels"):
        C.StemDepthtoChannels(1, 12, 96, (4, 4, 4), (2, 4, 4))


def test_scale_intensity_golden():
    """G4: the reference's BatchedRandScaleIntensity with its own draw (seed 11) == oracle with the drawn factors injected"""
    g = load_golden("intensity.pt")["scale"]
    assert torch.equal(transforms_ref.scale_intensity(g["x"], g["factors"]), g["y"])


def test_scale_intensity_channel_wise_golden_and_draw_order():
    """G4 channel_wise=True: oracle == reference output; and the product class draws what the reference drew (selection first,
    then one uniform per (sample, channel) — _scale_intensity.py:42-50) from the same seed of the global generator"""
    from viscy_amd.transforms import BatchedRandScaleIntensityd

    all_g = load_golden("intensity.pt")
    g = all_g["scale_channel_wise"]
    assert g["factors"].shape == (6, 2) and torch.equal(transforms_ref.scale_intensity(g["x"], g["factors"]), g["y"])
    t = BatchedRandScaleIntensityd(["a"], factors=g["range"], prob=g["prob"], channel_wise=True)
    torch.manual_seed(g["seed"])
    torch.testing.assert_close(t.randomize(6, 2), g["factors"], rtol=0, atol=1e-6)  # the golden holds (1 + f) - 1
    g1 = all_g["scale"]
    t1 = BatchedRandScaleIntensityd(["a"], factors=0.5, prob=0.5)
    torch.manual_seed(g1["seed"])
    torch.testing.assert_close(t1.randomize(6, 2), g1["factors"], rtol=0, atol=1e-6)


def test_oracle_adjust_contrast_options():
    """restated MONAI AdjustContrast options: retain_stats restores mean / std of the sample, invert_image mirrors the curve"""
    x = torch.rand((3, 2, 4, 8, 8), generator=torch.Generator().manual_seed(5)) * 4 - 1
    gamma, sel = torch.tensor([0.6, 1.0, 2.2]), torch.tensor([True, False, True])
    r = transforms_ref.adjust_contrast(x, gamma, sel, retain_stats=True)
    for i in (0, 2):
        assert abs(r[i].mean() - x[i].mean()) < 1e-5 and abs(r[i].std() - x[i].std()) < 1e-5 and not torch.allclose(r[i], x[i])
    assert torch.equal(r[1], x[1])
    inv = transforms_ref.adjust_contrast(x, gamma, sel, invert_image=True)
    torch.testing.assert_close(inv, -transforms_ref.adjust_contrast(-x, gamma, sel), rtol=0, atol=0)


def test_oracle_transforms_match_the_reference_pins():
    """G4b (oracle/validate_against_reference.py): outputs of the reference's own _noise.py / _flip.py / _crop.py (run on a
    stub of their MONAI base classes) with the draws the reference made; the oracle reproduces them from those draws."""
    from oracle import transforms_ref as R

    pins = load_golden("transform_pins.pt")
    for tag in ("noise_sampled_std", "noise_fixed_std"):
        p = pins[tag]
        y = R.gaussian_noise(p["x"], p["field"], p["std"], p["apply"], mean=p["mean"])
        torch.testing.assert_close(y, p["y"], rtol=0, atol=2e-6)
        assert torch.equal(y[~p["apply"]], p["x"][~p["apply"]]) and 0 < int(p["apply"].sum()) < p["x"].shape[0]
    p = pins["weighted_crop"]
    assert torch.equal(R.weighted_crop_window_weights(p["weight_map"], p["size"][1:]), p["weights"])
    assert torch.equal(R.crop3d(p["source"], p["z0"], p["y0"], p["x0"], p["size"]), p["source_out"])
    assert torch.equal(R.crop3d(p["weight_map"], p["z0"], p["y0"], p["x0"], p["size"]), p["target_out"])
    p = pins["flip"]
    assert p["flips"].any() and not p["flips"].all() and p["y"].shape == p["x"].shape
