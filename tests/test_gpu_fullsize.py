"""GPU: the configurations BASELINE.json quotes, at their FULL sizes, where the driver can see them.

* north-star gate shape — (B, 1, 5, 2048, 2048) bf16 training step: batch independence, linearity of the backward in the
  output gradient, hipGraph replay == eager launches (lr = 0), finite learning steps;
* config 4 — sliding-window prediction of a (1, 1, 21, 2048, 2048) FOV: graph-captured windows == eager windows, the
  device-side Z blend == the reference's `_blend_in` arithmetic (prediction_writer.py:74-111) applied to the SAME device
  window outputs, in fp32 (`32-true`, the reference's predict precision) and bf16.
The oracle cannot run these sizes in seconds, so parity is carried by size-independent properties here and by the
oracle / golden comparisons at the sizes it can (tests/test_gpu_model.py).
"""

import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4,
          decoder_conv_blocks=2)


def _model(dt, seed=42):
    import bench
    from viscy_amd.unext2 import UNeXt2

    torch.manual_seed(seed)
    m = UNeXt2(**KW).cuda()
    bench.nonzero_grn_(m)
    m.compute_dtype, m.grad_mode = dt, "flat"
    return m


def _free():
    import gc

    gc.collect()
    torch.cuda.empty_cache()


def test_gate_shape_training_step_properties_bf16():
    """(B = 8, Z = 5, 2048 x 2048) — the batch `bench.py`'s gate_shape record runs (the test ran B = 4): one sample is
    64 bench patches; the replay / learning parts take the full batch (~130 GB), the property parts sub-batches of it."""
    import bench
    from viscy_amd.losses import MixedLoss
    from viscy_amd.optim import FlatAdamW
    from viscy_amd.step import TrainStep

    m = _model(torch.bfloat16)
    eng = m.engine()
    x, t = bench.make_batch(8, 2048, 2048, "cuda", seed=5)
    # ---- batch independence of the forward (per-sample GRN / InstanceNorm statistics, tiles inside one sample)
    m.eval()
    with torch.no_grad():
        yb = m(x[:3])
        y1 = m(x[1:2])
    assert yb.shape == (3, 2, 5, 2048, 2048)
    assert ((yb[1:2] - y1).abs().max() / y1.abs().max()).item() < 2e-2
    assert torch.nn.functional.cosine_similarity(yb[1].flatten(), y1[0].flatten(), dim=0).item() > 0.9999
    del yb, y1
    _free()
    # ---- backward: linear in dout; batch gradient = sum of the per-sample gradients
    m.train()
    g = torch.Generator().manual_seed(6)
    dout = torch.randn((2, 2, 5, 2048, 2048), generator=g).cuda()

    def grad_of(xx, dd):
        eng.flat_grad.zero_()
        m(xx).backward(dd)
        return eng.flat_grad.clone()

    g1, g1b = grad_of(x[:2], dout), grad_of(x[:2], dout)
    g2 = grad_of(x[:2], 2.0 * dout)
    floor = torch.nn.functional.cosine_similarity(g1, g1b, dim=0).item()
    assert floor > 0.998
    assert torch.nn.functional.cosine_similarity(g1, g2, dim=0).item() > floor - 2e-3
    assert abs((g2.norm() / g1.norm()).item() - 2.0) < 4e-2
    gs = grad_of(x[:1], dout[:1]) + grad_of(x[1:2], dout[1:])
    assert torch.nn.functional.cosine_similarity(g1, gs, dim=0).item() > floor - 2e-3
    del g1, g1b, g2, gs, dout
    _free()
    # ---- hipGraph replay == eager launches at lr = 0 (same parameters -> same loss, same gradients)
    res = {}
    for mode in ("eager", "graph"):
        opt = FlatAdamW(eng, lr=0.0, weight_decay=0.0)
        step = TrainStep(m, MixedLoss(0.5, 0, 0.5), opt, use_graph=mode == "graph")
        losses = [float(step(x, t)) for _ in range(2)]
        res[mode] = (losses, eng.flat_grad.clone())
        del step, opt
        _free()
    (le, ge), (lg, gg) = res["eager"], res["graph"]
    assert abs(le[1] - lg[1]) < 2e-3 * abs(le[1])
    assert torch.nn.functional.cosine_similarity(ge, gg, dim=0).item() > 0.998
    # ---- learning at the gate shape: finite, decreasing
    opt = FlatAdamW(eng, lr=5e-4, schedule="WarmupCosine", warmup_steps=3, t_total=8, warmup_multiplier=1e-3)
    step = TrainStep(m, MixedLoss(0.5, 0, 0.5), opt, use_graph=True)
    ls = [float(step(x, t)) for _ in range(6)]
    assert all(torch.isfinite(torch.tensor(ls))) and ls[-1] < ls[0]
    assert torch.isfinite(eng.flat).all().item() and torch.isfinite(eng.flat_grad).all().item()


def _ref_blend(old, new, z_slice):  # prediction_writer.py:74-111 (torch branch), restated
    if z_slice.start == 0:
        return new
    depth = z_slice.stop - z_slice.start
    samples = min(z_slice.start + 1, depth)
    f = torch.tensor([min(i + 1, samples) for i in reversed(range(depth))], dtype=old.dtype, device=old.device).view(1, 1, -1, 1, 1)
    return old * (f - 1) / f + new / f


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32_32true", "bf16"])
def test_config4_full_fov_sliding_window_prediction(dt):
    """(1, 1, 21, 2048, 2048) -> (1, 2, 21, 2048, 2048): 17 windows of 5 slices, each a full 2048 x 2048 forward"""
    from viscy_amd.vsunet import VSUNet

    torch.manual_seed(0)
    vs = VSUNet("UNeXt2", dict(KW)).cuda().eval()
    import bench

    bench.nonzero_grn_(vs.model)
    vs.model.compute_dtype = dt
    g = torch.Generator().manual_seed(3)
    x = torch.nn.functional.avg_pool3d(torch.randn((1, 1, 21, 2048, 2048), generator=g), (1, 3, 3), 1, (0, 1, 1)).cuda()
    with torch.no_grad():
        eager = vs.predict_sliding_windows(x, out_channel=2)
        vs.predict_graph = True
        graphed = vs.predict_sliding_windows(x, out_channel=2)
        vs.predict_graph = False
        assert eager.shape == graphed.shape == (1, 2, 21, 2048, 2048) and torch.isfinite(eager).all().item()
        tol = 1e-3 if dt == torch.float32 else 2e-2
        assert ((graphed - eager).abs().max() / eager.abs().max()).item() < tol
        assert torch.nn.functional.cosine_similarity(graphed.flatten(), eager.flatten(), dim=0).item() > (0.999999 if dt == torch.float32 else 0.9999)
        del graphed
        # the Z blend: the reference's arithmetic over the SAME device window outputs (windows recomputed eagerly; fp32
        # windows are deterministic up to reduction order, so the comparison bar is the forward's own run-to-run floor)
        expect = torch.zeros_like(eager)
        for z0 in range(0, 21 - 5 + 1):
            zs = slice(z0, z0 + 5)
            pred = vs.predict_step({"source": x[:, :, zs].contiguous()}, 0)
            expect[:, :, zs] = _ref_blend(expect[:, :, zs], pred.float(), zs)
        assert ((eager - expect).abs().max() / expect.abs().max()).item() < tol
        # every output slice is a convex combination of window predictions: interior slices average 5 windows
        assert eager[:, :, 10].abs().max().item() > 0
