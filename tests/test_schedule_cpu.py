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


def test_bad_depth_and_cpu_forward_raise():
    with pytest.raises(ValueError, match="not divisible"):
        UNeXt2(in_stack_depth=7)
    m = UNeXt2(backbone="convnextv2_atto")
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 1, 5, 64, 64))
