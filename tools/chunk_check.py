"""Sample-chunk-major schedule (Engine._sample_chunk, VSX_CHUNK_MB) against the whole-batch schedule: same weights, same batch,
one bf16 forward + backward each; loss and the flat gradient must agree to reduction-order noise.  GPU only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from viscy_amd.losses import MixedLoss  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

B = int(os.environ.get("B", 64))
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4,
               decoder_conv_blocks=2).to(dev)
bench.nonzero_grn_(model)
model.compute_dtype, model.grad_mode = torch.bfloat16, "flat"
eng = model.engine()
crit = MixedLoss(0.5, 0.0, 0.5)
x, tgt = bench.make_batch(B, 256, 256, dev)
res = {}
for mb in ("0", "16"):
    os.environ["VSX_CHUNK_MB"] = mb
    for p in model.parameters():  # flat mode: the gradients are views of ONE buffer that a backward accumulates into
        if p.grad is not None:
            p.grad.zero_()
    y = model(x)
    loss = crit(y, tgt)
    loss.backward()
    torch.cuda.synchronize()
    g = torch.cat([p.grad.flatten().float() for p in model.parameters() if p.grad is not None])
    res[mb] = (float(loss), g.clone(), y.detach().float().clone())
    print("chunk MB", mb, "samples per chunk (64x64x224):", eng._sample_chunk(B, 4096, 224), "loss", float(loss), flush=True)
(l0, g0, y0), (l1, g1, y1) = res["0"], res["16"]
cos = float(torch.dot(g0, g1) / (g0.norm() * g1.norm()))
print("forward max diff / max", float((y0 - y1).abs().max() / y0.abs().max()), "loss", l0, l1, "grad cos", cos,
      "grad rel", float((g0 - g1).norm() / g0.norm()))
assert abs(l0 - l1) <= 2e-3 * abs(l0) and cos > 0.9995, "chunked schedule differs"
print("OK")
