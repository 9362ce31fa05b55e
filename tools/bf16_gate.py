"""How far is the bf16 engine (bf16 residual stream, fp32 accumulation) from the fp32 oracle — and how far is the
reference's OWN mixed-precision arithmetic (the oracle under ``torch.autocast(bfloat16)``: bf16 convolution / linear
inputs, fp32 LayerNorm and residual sums) from the same fp32 oracle?  Prints one JSON line per model case; the numbers set
the bars in ``tests/test_gpu_model.py::test_forward_backward_bf16_tracks_fp32``.

Also prints the per-element error distribution of the MixedLoss gradient against the oracle (for the per-element gate in
``test_mixed_loss_vs_reference_golden``).

Usage (GPU box): ``python tools/bf16_gate.py``.  Uses ``oracle/`` as the checker only.
"""

import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import loss_ref, unext2_ref  # noqa: E402


def cosine(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=0).item()


def model_case(tag, kw, bhw, seed=7):
    from viscy_amd.unext2 import UNeXt2

    ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=seed).eval()
    b, h, w = bhw
    x = torch.randn(b, kw["in_channels"], kw["in_stack_depth"], h, w, generator=torch.Generator().manual_seed(1))
    y = ref(x)
    dout = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(dout)
    g32 = torch.cat([p.grad.flatten() for p in ref.parameters()])
    per32 = {n: p.grad.clone() for n, p in ref.named_parameters()}
    ref.zero_grad()
    # the reference's mixed precision: autocast on the same module (CPU autocast = the same op-level cast policy)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ya = ref(x)
    ya.float().backward(dout)
    ga = torch.cat([p.grad.flatten() for p in ref.parameters()])
    pera = {n: p.grad.clone() for n, p in ref.named_parameters()}
    ref.zero_grad()
    mine = UNeXt2(**kw)
    mine.load_state_dict(ref.state_dict(), strict=True)
    mine = mine.cuda()
    mine.compute_dtype = torch.bfloat16
    out = mine(x.cuda())
    out.backward(dout.cuda())
    gm = torch.cat([p.grad.flatten().cpu() for p in mine.parameters()])
    perm = {n: p.grad.cpu() for n, p in mine.named_parameters()}
    ymax = y.detach().abs().max().item()

    def worst_param(per):
        w_, name = 1.0, None
        for n, g in per32.items():
            if g.abs().max() == 0:
                continue
            c = cosine(per[n].flatten(), g.flatten())
            if c < w_:
                w_, name = c, n
        return w_, name

    rec = {"case": tag, "fwd_err_hip_bf16": (out.cpu() - y.detach()).abs().max().item() / ymax,
           "fwd_err_ref_autocast": (ya.float().detach() - y.detach()).abs().max().item() / ymax,
           "fwd_rms_hip_bf16": ((out.cpu() - y.detach()).square().mean().sqrt() / y.detach().square().mean().sqrt()).item(),
           "fwd_rms_ref_autocast": ((ya.float().detach() - y.detach()).square().mean().sqrt() / y.detach().square().mean().sqrt()).item(),
           "grad_cos_hip_bf16": cosine(gm, g32), "grad_cos_ref_autocast": cosine(ga, g32),
           "grad_cos_hip_vs_autocast": cosine(gm, ga),
           "worst_param_cos_hip": worst_param(perm), "worst_param_cos_autocast": worst_param(pera)}
    print(json.dumps(rec), flush=True)


def loss_case(shape, seed, corr):
    from viscy_amd.losses import MixedLoss

    gen = torch.Generator().manual_seed(seed)
    target = torch.rand(shape, generator=gen)
    pred = target + 0.1 * torch.randn(shape, generator=gen) if corr else torch.rand(shape, generator=gen)
    p = pred.cuda().requires_grad_(True)
    MixedLoss(0.5, 0.0, 0.5)(p, target.cuda()).backward()
    pr = pred.clone().requires_grad_(True)
    loss_ref.mixed_loss(pr, target, 0.5, 0.0, 0.5).backward()
    g, gr = p.grad.cpu(), pr.grad
    d = (g - gr).abs() / gr.abs().max()
    q = torch.quantile(d.flatten()[:: max(1, d.numel() // 1_000_000)], torch.tensor([0.5, 0.99, 0.999]))
    print(json.dumps({"loss_case": [list(shape), seed, corr], "max_err_over_absmax": d.max().item(),
                      "median": q[0].item(), "p99": q[1].item(), "p999": q[2].item(),
                      "mean_abs_rel": ((g - gr).abs().mean() / gr.abs().mean()).item(), "cos": cosine(g.flatten(), gr.flatten())}),
          flush=True)


if __name__ == "__main__":
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    model_case("atto_pool", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_atto", head_pool=True),
               (2, 64, 96))
    model_case("tiny_pool", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True),
               (2, 128, 128))
    for shape, seed, corr in [((2, 2, 5, 192, 192), 0, False), ((2, 2, 5, 192, 192), 0, True), ((1, 2, 5, 256, 256), 3, True),
                              ((2, 1, 5, 176, 208), 5, True)]:
        loss_case(shape, seed, corr)
