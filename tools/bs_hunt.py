"""The bf16 engine of tests/test_gpu_model.py::test_baseline_size_* scored against the fp32 golden repeatedly in one process
(optionally with NaN-poisoned allocations: POISON=1; with the fp32 engine / an autocast oracle pass in between: MIX=1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unext2_ref  # noqa: E402  (test infrastructure: this is a tools/ script, not product code)
from tests.conftest import load_golden  # noqa: E402
from viscy_amd import debug  # noqa: E402
from viscy_amd.losses import MixedLoss  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

gold = load_golden("unext2_tiny_256.pt")
kw = gold["kwargs"]
B, S = gold["shape"]
ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=gold["seed"])
g = torch.Generator().manual_seed(gold["x_seed"])
x = torch.randn((B, 1, 5, S, S), generator=g)
with torch.no_grad():
    y0 = ref(x)
tgt = (y0 + gold["tgt_noise"] * y0.std() * torch.randn((B, 2, 5, S, S), generator=g)).contiguous()
st = gold["y_stride"]


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
    fwd = ((y.detach().float().cpu()[..., ::st, ::st] - gold["y"]).abs().max() / gold["y_absmax"]).item()
    out = {}
    for gname, names in gold["groups"].items():
        a, b = [], []
        for n in names:
            stride, sample = gold["grad_samples"][n]
            a.append(sample.double())
            b.append(eng.g(named[n]).flatten()[::stride].double().cpu())
        a, b = torch.cat(a), torch.cat(b)
        out[gname] = 1.0 - torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    return fwd, out


ctx = debug.poison_empty() if os.environ.get("POISON") else __import__("contextlib").nullcontext()
with ctx:
    for it in range(int(os.environ.get("N", 6))):
        if os.environ.get("MIX") and it % 2 == 1:
            run_engine(torch.float32)
            o = unext2_ref.UNeXt2(**kw).cuda()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o(x.cuda()).float().sum().backward()
        fwd, stg = run_engine(torch.bfloat16)
        print(f"it {it}: fwd {fwd:.2e} " + " ".join(f"{k.replace('enc_stages_', 's').replace('enc_', '')}={v:.1e}" for k, v in stg.items()), flush=True)
