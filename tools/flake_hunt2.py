"""Repeat the tiny_pool oracle-vs-HIP gradient comparison; on a miss, recompute both sides to see which one moved."""
import sys

import torch

sys.path.insert(0, ".")
from oracle import unext2_ref  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

CASES = [
    dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_atto", head_pool=True),
    dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True),
]
NAME = "encoder_stages.stages_0.blocks.0.norm.weight"


def grads_ref(ref, x, dout):
    for p in ref.parameters():
        p.grad = None
    y = ref(x)
    y.backward(dout)
    return {n: p.grad.clone() for n, p in ref.named_parameters()}


def grads_gpu(m, x, dout, mode):
    if mode == "flat":
        m.engine().flat_grad.zero_()
    else:
        for p in m.parameters():
            p.grad = None
    out = m(x.cuda())
    out.backward(dout.cuda())
    return {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for mode in ("autograd", "flat"):
        # a smaller model first, like the test session (different workspace sizes before the tiny case)
        for kw in CASES:
            torch.manual_seed(0)
            ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=7).eval()
            m = UNeXt2(**kw)
            m.load_state_dict(ref.state_dict(), strict=True)
            m = m.cuda()
            m.compute_dtype, m.grad_mode = torch.float32, mode
            x = torch.randn(2, 1, 5, 128, 128)
            dout = torch.randn_like(ref(x)).detach()
            gr = grads_ref(ref, x, dout)
            gg = grads_gpu(m, x, dout, mode)
            errs = sorted(((rel(gg[n], gr[n]), n) for n in gr if n != "head.conv.0.conv.bias"), reverse=True)
            print(it, mode, kw["backbone"], "top:", [(f"{e:.1e}", n) for e, n in errs[:3]], flush=True)
            if errs[0][0] > 1e-3:
                gr2 = grads_ref(ref, x, dout)
                gg2 = grads_gpu(m, x, dout, mode)
                n = errs[0][1]
                print("  MISS", n, "ref-vs-ref2", rel(gr2[n], gr[n]), "gpu-vs-gpu2", rel(gg2[n], gg[n]),
                      "gpu2-vs-ref", rel(gg2[n], gr[n]), "gpu-vs-ref2", rel(gg[n], gr2[n]), flush=True)
                d = (gg[n] - gr[n]).abs()
                print("  worst idx", d.argmax().item(), "gpu", gg[n].flatten()[d.argmax()].item(), "ref", gr[n].flatten()[d.argmax()].item(),
                      "n bad", (d > 1e-3 * gr[n].abs().max()).sum().item(), "of", d.numel(), flush=True)
