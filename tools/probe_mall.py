"""VERDICT r5 item 1a, kernel level: does a reader of the 4C-wide dh run faster when its chunk was written just before it (chunk
sized for the 256 MiB Infinity Cache) than when the chunk comes from HBM?  For chunks of S samples of the 64 x 64 maps:
  warm  = [write the dh chunk] -> reader, timed: the reader only
  cold  = [write the dh chunk] -> [stream 2 GiB through the caches] -> reader
  whole = the reader on the whole batch (B = 512), per-chunk share of its time
Readers: the fc1 weight gradient (TN) and the fc1 data gradient with LayerNorm backward (gemm_nt2_lnbwd).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viscy_amd import ops  # noqa: E402

B = int(os.environ.get("B", 512))
dt = torch.bfloat16
dev = "cuda"
flush = torch.empty(1 << 30, dtype=torch.int16, device=dev)  # 2 GiB


def timed(fn, pre, rep=7):
    ts = []
    for it in range(rep + 2):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for C, hw in ((224, 4096), (96, 4096)):
    M = B * hw
    dz = torch.empty((M, 4 * C), device=dev, dtype=dt)
    one = torch.randn((1, hw * 4 * C), device=dev, dtype=dt) * 0.1
    y = torch.randn((M, C), device=dev, dtype=dt)
    mean = torch.zeros(M, device=dev)
    rstd = torch.ones(M, device=dev)
    WT = torch.randn((C, 4 * C), device=dev, dtype=dt) * 0.05
    dW = torch.zeros((4 * C, C), device=dev)
    dy = torch.empty((M, C), device=dev, dtype=dt)
    dz.view(B, -1).copy_(one.expand(B, -1))

    def readers(r0, r1):
        Mc = r1 - r0
        return {
            "TN dW1": lambda: ops.gemm("tn", y[r0:r1], dz[r0:r1], dW, Mc, 4 * C, C, C, 4 * C, C, dtype=dt),
            "NT lnbwd": lambda: ops.dgrad_ln_bwd(dz[r0:r1], WT, y[r0:r1], rstd[r0:r1], Mc, C, 4 * C, mean=mean[r0:r1], out=dy[r0:r1]),
        }

    whole = {k: timed(f, lambda: None) for k, f in readers(0, M).items()}
    print(f"C = {C}: whole batch (B = {B}): " + ", ".join(f"{k} {v:8.1f} us" for k, v in whole.items()), flush=True)
    for S in (8, 16, 24, 32, 64, 128):
        r0, r1 = 3 * S * hw, 4 * S * hw
        mb = (r1 - r0) * 4 * C * 2 / 2**20

        def write():
            dz[r0:r1].view(S, -1).copy_(one.expand(S, -1))

        def write_flush():
            write()
            flush.zero_()

        line = f"  S = {S:3d} ({mb:6.1f} MiB):"
        for k, f in readers(r0, r1).items():
            w, c = timed(f, write), timed(f, write_flush)
            line += f"  {k}: warm {w:7.1f} cold {c:7.1f} us (warm/cold {w / c:4.2f}; whole-batch share {whole[k] * S / B:7.1f})"
        print(line, flush=True)
    del dz, y, dy
